"""Parity of the exact launch form bench.py times: the event stream resident in HBM, the
step reading its batch at a device-side offset and advancing it, ONE captured hipGraph replayed per batch, lean
direct-form eager step with pre-multiplied attention weights and the row bound that selects the small updater
blocks - against the oracle after every replay, and the final state.  C3: the lazy-restart triggers fire INSIDE
the replayed graph (train_self_supervised.py:152-163 with the static restarter, restarters.py:254-277)."""
import os

import numpy as np
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device('cuda', 0)


def _resident(stream):
    return tuple(torch.from_numpy(stream[k]).to(dev()) for k in ('src', 'dst', 'neg', 'ts', 'eids'))


def _capture(model, buf, extra=()):
    """bench.py::run_stream_leg's capture: side stream, offset (and lazy batch counter) restored afterwards"""
    side = torch.cuda.Stream()
    snap = [t.clone() for t in (buf.offset,) + tuple(extra)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        model.launch_step(buf)
    for t, s in zip((buf.offset,) + tuple(extra), snap):
        t.copy_(s)
    torch.cuda.synchronize()
    return graph


@pytest.mark.parametrize('prefetch', [False, True], ids=['collate_in_step', 'collate_prefetched'])
@pytest.mark.parametrize('n_eager', [3, 0], ids=['after_eager_steps', 'first_step_replayed'])
def test_c2_graph_replay_of_the_resident_lean_step_matches_oracle(n_eager, prefetch):
    """BASELINE configs[1] in bench.py's timed form: n_eager eager steps (state pre-roll; they also give the model the
    row bound that selects k_gru_direct), then the captured step replayed 12 times - embeddings compared after every
    replay, memories / mailbox / has-message set at the end."""
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    c = bench.C2
    B, K, d = c['B'], c['K'], c['d']
    nb = n_eager + 12
    E = (nb + 2) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=21, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], with_oracle=True)
    model.fuse_attention()
    model.eager_updates()
    # prefetch (bench.py's default): sampler + centres of the next batch ride on the step's last launch (tg_step_io.
    # prefetch_state); the neighbour lists are then not an output of the step
    buf = model.StepBuffers(model, B, False, resident=_resident(stream), prefetch=prefetch, debug_lists=prefetch)
    buf.io.lean = 1
    _ = model.graph.tcsr, model.model_struct()

    def check(b):
        torch.cuda.synchronize()
        assert int(buf.err.item()) == 0
        assert int(buf.offset.item()) == (b + 1) * B
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        if not prefetch:
            np.testing.assert_array_equal(buf.l1_nids.cpu().numpy(), cg['l1_nids'])
        else:  # the lists the step consumed (prefetched by the previous step's last launch): bit-exact (graph.py:67-148)
            np.testing.assert_array_equal(buf.dbg_l1_nids.cpu().numpy(), cg['l1_nids'])
            np.testing.assert_array_equal(buf.dbg_l1_eids.cpu().numpy(), cg['l1_eids'])
            np.testing.assert_array_equal(buf.dbg_l1_ts.cpu().numpy(), cg['l1_ts'])
        if not prefetch:
            pass
        elif b >= 1 and n_eager and os.environ.get('TG_PREFETCH', '1') != '0' and os.environ.get('TG_GTAB', '1') != '0':
            # (a graph captured at the very first step replays collate + prefetch: the flag stays 0; the knobs switch it off)
            assert buf._pf_state.value == 1  # every step after the first started with its attention core
        cnt = buf.counts.tolist()
        assert cnt[0] == -1 and cnt[2] == len(cg['rd_nids'])  # lean form taken; unique positives still counted
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, f'h_left, batch {b}', TOL)
        return cnt

    for b in range(n_eager):
        model.launch_step(buf)
        cnt = check(b)
        model.note_rows(cnt[1], cnt[2])
    if n_eager == 0:  # the derived tables must be current before a capture (their rebuilds refuse to run inside one)
        model._sync_pending()
        model._sync_gtab()
    graph = _capture(model, buf)
    for b in range(n_eager, nb):
        graph.replay()
        check(b)
    compare_state_with_oracle(model, orc)


def test_c2_graph_of_several_steps_matches_oracle():
    """bench.py's timed region replays graphs of SEVERAL steps (consecutive replays are ~9 us apart on the device): four
    steps per captured graph here, with the collate prefetch chaining them inside the graph; the embeddings of the last
    step of every replay and the final state against the oracle."""
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    c = bench.C2
    B, K, d = c['B'], c['K'], c['d']
    n_eager, gsteps, n_replays = 3, 4, 3
    nb = n_eager + gsteps * n_replays
    E = (nb + 2) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=37, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], with_oracle=True)
    model.fuse_attention()
    model.eager_updates()
    buf = model.StepBuffers(model, B, False, resident=_resident(stream), prefetch=True, debug_lists=True)
    buf.io.lean = 1
    _ = model.graph.tcsr, model.model_struct()
    last_cg = {}

    def oracle_step(b):
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        last_cg.update(cg)
        return orc.stream_step(*a, cg).numpy()

    for b in range(n_eager):
        model.launch_step(buf)
        torch.cuda.synchronize()
        ref = oracle_step(b)
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
        assert int(buf.err.item()) == 0 and int(buf.offset.item()) == b * B
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, f'h_left, replay {r}', TOL)
        # the neighbour lists the LAST step of the replay consumed (prefetched inside the graph): bit-exact
        np.testing.assert_array_equal(buf.dbg_l1_nids.cpu().numpy(), last_cg['l1_nids'])
        np.testing.assert_array_equal(buf.dbg_l1_eids.cpu().numpy(), last_cg['l1_eids'])
        np.testing.assert_array_equal(buf.dbg_l1_ts.cpu().numpy(), last_cg['l1_ts'])
    compare_state_with_oracle(model, orc)


@pytest.mark.parametrize('lean', [False, True], ids=['full_step', 'lean_tables'])
def test_c3_graph_replay_with_lazy_restart_triggers_inside_matches_reference_loop(lean):
    """BASELINE configs[2] as bench.py --workload c3 times it: static restarter, the restart draws made up front, the
    forget / re-initialise loop inside the step and the step inside a replayed graph; two triggers fall into the
    replayed region.  Oracle: the reference's loop on the host."""
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    c = bench.WORKLOADS['c3']
    B, K, d = c['B'], c['K'], c['d']
    n_eager, nb = 2, 9
    E = (nb + 1) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=23, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], restarter='static', with_oracle=True)
    torch.manual_seed(12)
    with torch.no_grad():
        for nm in ('left_emb', 'right_emb'):
            tbl = getattr(model.restarter_fn, nm).weight
            tbl.normal_(0.0, 0.5)
            orc.p[f'restarter_fn.{nm}.weight'] = tbl.detach().cpu().clone()
    model.fuse_attention()
    model.eager_updates()
    trigger = np.zeros(nb, dtype=np.uint8)
    trigger[[3, 6]] = 1
    buf = model.StepBuffers(model, B, False, resident=_resident(stream))
    buf.enable_lazy_restart(model, trigger)
    # lean (bench.py --workload c3's form): the involved flags are marked for the loop but no sorted set is formed, and the
    # per-node query-row / centre-row tables stay in use - the loop's kernel rewrites the rows of what it re-initialises
    buf.io.lean = 1 if lean else 0
    _ = model.graph.tcsr, model.model_struct()
    restarting, uptodate, total = False, set(), 0
    graph = None
    for b in range(nb):
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        n_r = 0
        if trigger[b] and b:
            restarting, uptodate = True, set()
            orc.clear_msgs()
        if restarting:
            r = np.array(sorted(set(cg['involved'].tolist()) - uptodate), dtype=np.int64)
            orc.restart(r, np.full(len(r), np.float32(a[3].min()), dtype=np.float32))
            uptodate.update(r.tolist())
            n_r = len(r)
        if b == n_eager:
            graph = _capture(model, buf, extra=(buf.lazy_batch,))
        if graph is not None:
            graph.replay()
        else:
            model.launch_step(buf)
        torch.cuda.synchronize()
        assert int(buf.err.item()) == 0
        ref = orc.stream_step(*a, cg).numpy()
        cnt = buf.counts.tolist()
        assert cnt[0] == (-1 if lean else len(cg['involved'])) and cnt[3] == n_r, (b, cnt, n_r)
        np.testing.assert_array_equal(buf.l1_nids.cpu().numpy(), cg['l1_nids'])
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, f'h_left, batch {b}', TOL)
        if graph is None:
            model.note_rows(cnt[1], cnt[2])
        total += n_r
    assert total > 5000
    compare_state_with_oracle(model, orc)


def test_graph_replay_and_eager_launches_are_bit_identical():
    """The same C2 batches through a captured graph (model A) and through eager launches (model B, same weights): every
    state tensor and the embeddings must be EQUAL, not close.  The updater's rows arrive in the order of an atomic
    compaction, which differs from run to run: no result may depend on a row's place in its tile (k-group partial sums
    are folded in group order, stream-K pieces in worker order)."""
    import bench
    c = bench.C2
    B, K, d = c['B'], c['K'], c['d']
    n_eager, n_graph = 3, 9
    E = (n_eager + n_graph + 2) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=29, d_e=d)
    res = _resident(stream)
    models, bufs = [], []
    for _ in range(2):
        m, _ = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'])
        m.fuse_attention()
        m.eager_updates()
        b = m.StepBuffers(m, B, False, resident=res)
        b.io.lean = 1
        models.append(m)
        bufs.append(b)
    for _ in range(n_eager):
        for m, b in zip(models, bufs):
            m.launch_step(b)
    torch.cuda.synchronize()
    for m, b in zip(models, bufs):
        cnt = b.counts.tolist()
        m.note_rows(cnt[1], cnt[2])
    graph = _capture(models[0], bufs[0])
    for step in range(n_graph):
        graph.replay()
        models[1].launch_step(bufs[1])
        torch.cuda.synchronize()
        assert torch.equal(bufs[0].h, bufs[1].h), f'embeddings differ at replay {step}'
    a, b = models
    for name, x, y in [('left', a.left_memory.vals, b.left_memory.vals), ('right', a.right_memory.vals, b.right_memory.vals),
                       ('left ts', a.left_memory.update_ts, b.left_memory.update_ts),
                       ('right ts', a.right_memory.update_ts, b.right_memory.update_ts),
                       ('pending', a._pending, b._pending), ('mailbox', a.msg_store.node_msg_vals, b.msg_store.node_msg_vals),
                       ('mailbox ts', a.msg_store.node_msg_ts, b.msg_store.node_msg_ts),
                       ('has_msg', a.msg_store.has_msg_bits, b.msg_store.has_msg_bits)]:
        assert torch.equal(x, y), f'{name} differs between graph replay and eager launches'


def test_prefetched_collate_is_discarded_when_state_offset_or_form_change():
    """tg_step_io.prefetch_state: the next batch's sampler + centres run on the step's last launch.  Whatever invalidates
    that work between two steps - state written outside the step (flush_msg, restart), a step of another form in between
    (full step: involved set formed), the stream offset moved, another buffer's step - must make the next step discard it
    (dedup slots cleared) and collate itself; the embeddings and the final state follow the oracle throughout."""
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    c = bench.C2
    B, K, d = 256, c['K'], c['d']
    nb = 14
    E = (nb + 2) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=31, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], with_oracle=True)
    model.fuse_attention()
    model.eager_updates()
    res = _resident(stream)
    buf = model.StepBuffers(model, B, False, resident=res, prefetch=True)
    buf.io.lean = 1
    other = model.StepBuffers(model, B, False, resident=res)  # a second buffer on the same model (no prefetch)
    other.io.lean = 1
    used = []

    def step(b, which=buf):
        which.offset.fill_(b * B)
        model.launch_step(which)
        torch.cuda.synchronize()
        assert int(which.err.item()) == 0
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        assert_close(which.h[:2 * B].cpu().numpy(), ref, f'h_left, batch {b}', TOL)

    def run(b):  # consecutive batch on the prefetching buffer WITHOUT touching the offset tensor
        before = buf._pf_state.value
        model.launch_step(buf)
        torch.cuda.synchronize()
        used.append(before)
        assert int(buf.err.item()) == 0 and int(buf.offset.item()) == (b + 1) * B
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, f'h_left, batch {b}', TOL)

    step(0)
    run(1); run(2)
    model.flush_msg(); orc.flush_msg()           # state written outside the step
    run(3); run(4)
    step(5, other)                               # another buffer's step on the same model
    buf.offset.fill_(6 * B)
    run(6); run(7)
    buf.io.lean = 0                              # a full step (involved set formed): cannot use the prefetch
    run(8)
    buf.io.lean = 1
    run(9); run(10)
    r = np.unique(stream['src'][11 * B:12 * B])[:40].astype(np.int64)   # TIGER.restart on some nodes of the next batch
    rt = np.full(len(r), float(np.float32(stream['ts'][11 * B])), dtype=np.float32)
    model.restart(torch.from_numpy(r).to(dev()), torch.from_numpy(rt).to(dev())); orc.restart(r, rt)
    run(11); run(12); run(13)
    # batches 2, 4, 7, 10, 12, 13 started from a valid prefetch; 3 (flush), 6 (other buffer + offset), 8 (full step),
    # 9 (after the full step: nothing was prefetched), 11 (restart) did not
    assert model._step_serial == 14
    if os.environ.get('TG_PREFETCH', '1') != '0' and os.environ.get('TG_GTAB', '1') != '0':  # (knobs that switch it off)
        assert used.count(1) >= 6
    compare_state_with_oracle(model, orc)


@pytest.mark.parametrize('side', ['1', '0'], ids=['side_stream', 'one_stream'])
def test_large_batch_side_stream_form_matches_oracle(side, monkeypatch):
    """Batches above 16 384 events host no riders; with TG_SIDE_STREAM=1 the write-back (STEP 4-5) runs on the library's
    side stream beside fc1 and the NEXT batch's sampler beside the updater and the query rows (csrc/tg_model.hip:
    SideLane), as parallel branches of a captured graph too (opt-in: measured not faster).  B = 20 000 on a small graph:
    two eager steps, then a graph of two steps replayed twice; neighbour lists bit-exact, embeddings and the final
    state against the oracle - in both forms."""
    monkeypatch.setenv('TG_SIDE_STREAM', side)
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    B, K, d = 20000, 10, 32
    n_eager, gsteps, n_replays = 2, 2, 2
    nb = n_eager + gsteps * n_replays
    E = (nb + 1) * B
    stream = bench.make_stream(30000, 8000, E, 4.0e5, seed=11, d_e=d)
    model, orc = bench.build_models(stream, d, K, 'left', 'left', with_oracle=True)
    model.fuse_attention()
    model.eager_updates()
    buf = model.StepBuffers(model, B, False, resident=_resident(stream), prefetch=True, debug_lists=True)
    buf.io.lean = 1
    _ = model.graph.tcsr, model.model_struct()
    side_on = side == '1' and os.environ.get('TG_PREFETCH', '1') != '0'

    def oracle_step(b, compare):
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        if compare:
            torch.cuda.synchronize()
            assert int(buf.err.item()) == 0 and int(buf.offset.item()) == (b + 1) * B
            np.testing.assert_array_equal(buf.dbg_l1_nids.cpu().numpy(), cg['l1_nids'])
            np.testing.assert_array_equal(buf.dbg_l1_eids.cpu().numpy(), cg['l1_eids'])
            np.testing.assert_array_equal(buf.dbg_l1_ts.cpu().numpy(), cg['l1_ts'])
            cnt = buf.counts.tolist()
            assert cnt[0] == -1 and cnt[2] == len(cg['rd_nids'])
            assert_close(buf.h[:2 * B].cpu().numpy(), ref, f'h_left, batch {b}', TOL)
            return cnt

    for b in range(n_eager):
        model.launch_step(buf)
        cnt = oracle_step(b, True)
        model.note_rows(cnt[1], cnt[2])
        if b >= 1:  # side stream: the step started with its attention core, its sampler ran beside the last updater
            assert buf._pf_state.value == (1 if side_on else 0)
    side = torch.cuda.Stream()
    snap = buf.offset.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(gsteps):
            model.launch_step(buf)
    buf.offset.copy_(snap)
    torch.cuda.synchronize()
    b = n_eager
    for _ in range(n_replays):
        graph.replay()
        for j in range(gsteps):
            oracle_step(b, j == gsteps - 1)
            b += 1
    compare_state_with_oracle(model, orc)
