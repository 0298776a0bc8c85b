"""Long-horizon agreement of the two most different forms of the step on one stream (both on the GPU): the round-1 form
(lazy updater, attention weights as stored, involved set formed) against the benchmarked form (eager updates in the
direct form, pre-multiplied weights, lean).  python tools/long_soak.py [n_batches [workload]]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
import bench
from _util import row_rel_err
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
c = bench.WORKLOADS[sys.argv[2]] if len(sys.argv) > 2 else bench.C2
B = c['B']
E = n * B
nf = bool(c.get('no_feats'))
stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=11, d_e=c['d'],
                           integer_ts=c.get('integer_ts', True), with_efeats=not nf)
a, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'], zero_nfeats=not nf)
b, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'], zero_nfeats=not nf)
b.fuse_attention(); b.eager_updates()
# ... and the timed form of bench.py: the stream resident in HBM, the collate prefetch (next batch's sampler + centres on the
# step's last launch), per-node query-row / centre-row tables, write-back riders
p, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'], zero_nfeats=not nf)
p.fuse_attention(); p.eager_updates()
res = tuple(torch.from_numpy(stream[k]).to(p.device) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
pbuf = p.StepBuffers(p, B, False, resident=res, prefetch=True)
pbuf.io.lean = 1
worst = 0.0
for i in range(n):
    s = [stream[k][i * B:(i + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
    ha = a.stream_step(*s, check_invariants=(i % 50 == 0))
    hb = b.stream_step(*s, check_invariants=(i % 50 == 0), lean=True)
    p.launch_step(pbuf)
    if i % 50 == 0:
        cnt = pbuf.counts.tolist()
        p.note_rows(cnt[1], cnt[2])
        assert int(pbuf.err.item()) == 0
    if i % 100 == 0 or i == n - 1:
        y = ha.h[:2 * B].cpu().numpy()
        for tag, hx in (('eager lean', hb.h), ('resident + prefetch', pbuf.h)):
            x = hx[:2 * B].cpu().numpy()
            e = float(np.abs(x - y).max() / max(1.0, np.abs(y).max())); r = row_rel_err(x, y)
            worst = max(worst, e, r)
            print(i, tag, 'h: max-abs-rel %.2e row-rel %.2e' % (e, r), flush=True)
for m2, tag in ((b, 'eager lean'), (p, 'resident + prefetch')):
    for nm in ('left_memory', 'right_memory'):
        x, y = getattr(m2, nm).vals.cpu().numpy(), getattr(a, nm).vals.cpu().numpy()
        e = float(np.abs(x - y).max() / max(1.0, np.abs(y).max()))
        worst = max(worst, e)
        print(tag, nm, '%.2e' % e, 'ts equal', bool(torch.equal(getattr(m2, nm).update_ts, getattr(a, nm).update_ts)))
    print(tag, 'has_msg equal', bool(torch.equal(a.msg_store.has_msg_bits, m2.msg_store.has_msg_bits)))
print('worst %.2e' % worst)
assert worst < 1e-4
