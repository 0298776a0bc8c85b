"""One C2 timed-form parity run in a process of its own (the library reads its knobs once per process): resident stream,
lean eager step with pre-multiplied weights and per-node tables, collate prefetch, a captured graph of three steps replayed
twice - embeddings, neighbour lists of the consumed batch and the final state against the oracle.  Run by
tests/test_hip_knobs.py with one non-default knob setting in the environment; prints `KNOB-CASE OK ...` on success."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import bench  # noqa: E402
from _util import assert_close  # noqa: E402
from oracle import tiger_oracle as O  # noqa: E402
from test_hip_parity import compare_state_with_oracle  # noqa: E402

TOL = 1e-4


def main():
    c = bench.C2
    B, K, d = c['B'], c['K'], c['d']
    n_eager, gsteps, n_replays = 2, 3, 2
    nb = n_eager + gsteps * n_replays
    E = (nb + 2) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=41, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], with_oracle=True)
    model.fuse_attention()
    model.eager_updates()
    dev = model.device
    res = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
    pf = os.environ.get('TG_KNOB_CASE_PREFETCH', '1') != '0'
    buf = model.StepBuffers(model, B, False, resident=res, prefetch=pf, debug_lists=True)
    buf.io.lean = 1
    _ = model.graph.tcsr, model.model_struct()
    last = {}

    def oracle_step(b):
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        last.clear()
        last.update(cg)
        return orc.stream_step(*a, cg).numpy()

    def check(b, what):
        assert int(buf.err.item()) == 0 and int(buf.offset.item()) == b * B
        np.testing.assert_array_equal(buf.dbg_l1_nids.cpu().numpy(), last['l1_nids'])
        np.testing.assert_array_equal(buf.dbg_l1_eids.cpu().numpy(), last['l1_eids'])
        return assert_close(buf.h[:2 * B].cpu().numpy(), ref, what, TOL)

    worst = 0.0
    for b in range(n_eager):
        model.launch_step(buf)
        torch.cuda.synchronize()
        ref = oracle_step(b)
        worst = max(worst, *check(b + 1, f'h_left, eager step {b}'))
        cnt = buf.counts.tolist()
        model.note_rows(cnt[1], cnt[2])
    side = torch.cuda.Stream()
    snap = buf.offset.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(gsteps):
            model.launch_step(buf)
    buf.offset.copy_(snap)
    torch.cuda.synchronize()
    b = n_eager
    for r in range(n_replays):
        graph.replay()
        torch.cuda.synchronize()
        for _ in range(gsteps):
            ref = oracle_step(b)
            b += 1
        worst = max(worst, *check(b, f'h_left, replay {r}'))
    compare_state_with_oracle(model, orc)
    print(f'KNOB-CASE OK worst {worst:.2e}', flush=True)


if __name__ == '__main__':
    main()
