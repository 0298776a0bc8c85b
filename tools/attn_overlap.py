#!/usr/bin/env python3
"""Experiment: the attention block (centres -> q.g product -> gather/softmax core -> fc1 -> fc2) of one C2 batch as
ONE chain over Q = 3B centres versus P independent chains of Q / P centres on P streams (fork / join inside one
captured graph).  The chains of different centres are independent; the question is whether the matrix-bound
products of one chain overlap with the gather-bound core of another."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from www2023tiger_amd import hip_ops  # noqa: E402
from www2023tiger_amd._lib import check, lib, ptr  # noqa: E402
from www2023tiger_amd.data.data_loader import GraphCollator  # noqa: E402

c = bench.C2
B = c['B']
E = 40 * B
st = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter='static')
model.fuse_attention()
model.eval()
dev = model.device
for b in range(20):
    sl = slice(b * B, (b + 1) * B)
    model.stream_step(*(st[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
sl = slice(20 * B, 21 * B)
coll = GraphCollator(model.graph, c['K'], 1, restarter='static', hist_len=1)
src, dst, neg, ts, eids, _, cg = coll.collate_arrays(*(st[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
m = model.model_struct()
err = hip_ops.new_err(dev)
cap = 3 * B * (c['K'] + 1)
comp, reprs = model._consume(cg.bitmap, cap, err)
ids = torch.cat([src, dst, neg]).long().to(dev).contiguous()
ts3 = ts.float().to(dev).repeat(3).contiguous()
l1_n, l1_e, l1_t = (x.contiguous() for x in cg.layers[1])
Q, d = ids.numel(), c['d']


def make_chain(lo, hi):
    q = hi - lo
    out = torch.empty(q, d, device=dev)
    nbytes = int(lib.tg_temporal_attn_workspace_bytes(C.byref(m), q))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a = (ids[lo:hi].contiguous(), ts3[lo:hi].contiguous(), l1_n[lo:hi].contiguous(), l1_e[lo:hi].contiguous(),
         l1_t[lo:hi].contiguous())

    def run():
        check(lib.tg_temporal_attn_fwd(C.byref(m), q, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(a[4]), ptr(reprs),
                                       ptr(cg.bitmap), ptr(comp['rank']), ptr(out), ptr(ws), nbytes,
                                       hip_ops.stream_ptr(dev)), 'attn')
    return run, out, (ws, a)


def timed(P):
    bounds = [Q * i // P for i in range(P + 1)]
    chains = [make_chain(bounds[i], bounds[i + 1]) for i in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    for run, _, _ in chains:
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap_stream = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=cap_stream):
        for rep in range(10):
            cur = torch.cuda.current_stream()
            for s_, (run, _, _) in zip(streams, chains):
                s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    run()
            for s_ in streams:
                cur.wait_stream(s_)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 50, torch.cat([o for _, o, _ in chains])


ref_t, ref = timed(1)
print(f'1 chain  of {Q} centres: {ref_t:.1f} us')
for P in (2, 3, 4, 6):
    t, out = timed(P)
    print(f'{P} chains of {Q // P} centres: {t:.1f} us   max|diff| vs one chain {float((out - ref).abs().max()):.2e}', flush=True)
