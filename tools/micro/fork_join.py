#!/usr/bin/env python3
"""Experiment: what does ONE fork / join inside a captured hipGraph cost?  Main chain A -> B -> D of matrix products; C is
independent of A and B and needed by D.  Serial (A B C D on one stream) against forked (C on a side stream, forked before A,
joined before D), both captured and replayed."""
import torch

dev = torch.device('cuda')
def mk(n): return torch.randn(n, n, device=dev), torch.randn(n, n, device=dev), torch.empty(n, n, device=dev)
for nA, nC in ((1024, 768), (1536, 768), (1024, 1024)):
    a, b, d, c = mk(nA), mk(nA), mk(nA), mk(nC)
    def A(): torch.mm(a[0], a[1], out=a[2])
    def B(): torch.mm(a[2], b[1], out=b[2])
    def Cc(): torch.mm(c[0], c[1], out=c[2])
    def D(): torch.mm(b[2], d[1], out=d[2])
    def timed(fn, reps=20):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (5 * reps)
    side = torch.cuda.Stream()
    def serial(): A(); B(); Cc(); D()
    def no_c(): A(); B(); D()
    def only_c(): Cc()
    def forked():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            Cc()
        A(); B()
        cur.wait_stream(side)
        D()
    print(f'n={nA} side n={nC}: A B D {timed(no_c):.1f} us | C alone {timed(only_c):.1f} | serial A B C D {timed(serial):.1f} | C forked {timed(forked):.1f}', flush=True)
