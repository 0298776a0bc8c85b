"""Multi-GPU streaming: one process per GPU, events of a batch sharded by the owner of
their destination node, node state replicated and kept coherent by ONE collective per
batch (an all-gather of the positive nodes' new rows over RCCL/xGMI).

Why this shape (SURVEY.md s8 e): every rank embeds only its shard of the global batch
(sampling, GRU, attention - the expensive part), reading a replica of the memories that
is exact because the exchange period is 1.  The write-back (STEP 4-6) is cheap, needs
only the 2 rows per event that the all-gather delivers, and is applied redundantly on
every rank, so no second collective and no remote reads are needed.  An all-gather
drives all 7 xGMI links of a GPU at once (a ring all-reduce would be bound by one link).
Results are bit-identical to the single-GPU engine run on the same global batch.
All state fits replicated for every BASELINE config (C5: 67 GB of 288 GB per GPU);
partitioning the tables themselves (all-to-all of remote rows) is future work.

The reference's own multi-GPU mode (time-chunk DDP, train_self_supervised_ddp.py) is a
different algorithm and is not what this file implements.
"""
import ctypes as C
import json
import os
import time
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as tdist


# --------------------------------------------------------------------------- host logic
def balanced_owner_table(n_nodes: int, dst: np.ndarray, world: int) -> np.ndarray:
    """owner[node] in [0, world).  dst-hash sharding with the hash chosen as a static
    lookup table that balances load: destination nodes are dealt to ranks in order of
    decreasing popularity (longest-processing-time first), so that a few very hot items
    (JODIE item popularity is Zipf-like) do not pile up on one rank.  Nodes that never
    occur as a destination fall back to node % world."""
    deg = np.bincount(dst, minlength=n_nodes).astype(np.int64)
    owner = (np.arange(n_nodes) % world).astype(np.int64)
    load = np.zeros(world, dtype=np.int64)
    for node in np.argsort(-deg, kind='stable'):
        if deg[node] == 0:
            break
        r = int(np.argmin(load))
        owner[node] = r
        load[r] += deg[node]
    return owner


class ShardPlan:
    """Where every event of one global batch is embedded and where its rows land in the
    gathered tensor G[world, kind(0 = h_left, 1 = h_new), role(0 = src, 1 = dst), cap, d]."""

    def __init__(self, dst: np.ndarray, owner: np.ndarray, world: int, cap: int):
        Bg = len(dst)
        self.Bg, self.world, self.cap = Bg, world, cap
        rank_of = owner[dst]
        self.counts = np.bincount(rank_of, minlength=world).astype(np.int64)
        if self.counts.max() > cap:
            raise ValueError(f'shard of {self.counts.max()} events exceeds capacity {cap}')
        order = np.argsort(rank_of, kind='stable')  # events grouped by rank, stream order kept inside a rank
        starts = np.concatenate([[0], np.cumsum(self.counts)[:-1]])
        slot = np.empty(Bg, dtype=np.int64)
        slot[order] = np.arange(Bg) - np.repeat(starts, self.counts)
        self.local_idx = [order[starts[r]:starts[r] + self.counts[r]] for r in range(world)]
        role = np.repeat(np.array([0, 1]), Bg)
        r2, s2 = np.tile(rank_of, 2), np.tile(slot, 2)
        self.left_row = ((r2 * 2 + 0) * 2 + role) * cap + s2   # [2Bg] row of h_left for position i of cat[src,dst]
        self.new_row = ((r2 * 2 + 1) * 2 + role) * cap + s2    # [2Bg] row of h(t'+)


def all_gather_rows(send: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """send [2, 2, cap, d] -> [world, 2, 2, cap, d].  On RCCL the gather runs on device;
    gloo (CPU tests, or two ranks sharing one GPU) stages through host memory."""
    out = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
    if tdist.get_backend(group) == 'nccl':
        tdist.all_gather_into_tensor(out, send.contiguous(), group=group)
        return out
    parts = [torch.empty(send.shape, dtype=send.dtype) for _ in range(world)]
    tdist.all_gather(parts, send.detach().cpu().contiguous(), group=group)
    return torch.stack(parts).to(send.device)


class ShardedRunner:
    """Drives one global batch: local embed -> all-gather -> replicated write-back.
    `backend` supplies the two compute halves (HipBackend in production; the CPU tests
    plug in an oracle-backed object to exercise this host logic under gloo)."""

    def __init__(self, backend, owner: np.ndarray, rank: int, world: int, cap: int, group=None):
        self.backend, self.owner, self.rank, self.world, self.cap, self.group = backend, owner, rank, world, cap, group

    def plan(self, dst: np.ndarray) -> ShardPlan:
        return ShardPlan(np.asarray(dst), self.owner, self.world, self.cap)

    def step(self, src, dst, neg, ts, eids, plan: Optional[ShardPlan] = None):
        """Arrays of the GLOBAL batch (host numpy).  Returns this rank's local embeddings."""
        plan = plan or self.plan(dst)
        li = plan.local_idx[self.rank]
        h_left, h_new = self.backend.embed(src[li], dst[li], neg[li], ts[li], eids[li])  # [2n, d] each
        n, d = len(li), h_left.shape[1]
        send = torch.zeros(2, 2, self.cap, d, dtype=torch.float32, device=h_left.device)
        send[0, 0, :n], send[0, 1, :n] = h_left[:n], h_left[n:2 * n]
        send[1, 0, :n], send[1, 1, :n] = h_new[:n], h_new[n:2 * n]
        gathered = all_gather_rows(send, self.world, self.group)
        self.backend.writeback(src, dst, ts, eids, gathered.reshape(-1, d), plan.left_row, plan.new_row)
        return h_left


class HipBackend:
    """The two halves on the HIP engine: tg_stream_step(embed_only) and the row-indexed
    write-back entry points."""

    def __init__(self, model, cap: int):
        from . import hip_ops
        from ._lib import check, lib, ptr
        self.model, self.cap = model, cap
        self.hip_ops, self.check, self.lib, self.ptr = hip_ops, check, lib, ptr
        self.buf = model.StepBuffers(model, cap, False, embed_only=True)
        self.err = hip_ops.new_err(model.device)

    def _dev(self, a, dt):
        return torch.as_tensor(a).to(self.model.device, dt).contiguous()

    def embed(self, src, dst, neg, ts, eids):
        n, buf = len(src), self.buf
        if n == 0:
            d = self.model.memory_dim
            z = torch.zeros(0, d, device=self.model.device)
            return z, z
        buf.src[:n], buf.dst[:n], buf.neg[:n] = (self._dev(x, torch.int64) for x in (src, dst, neg))
        buf.ts[:n], buf.eids[:n] = self._dev(ts, torch.float64), self._dev(eids, torch.int64)
        buf.io.B = n
        self.model.launch_step(buf)
        return buf.h[:2 * n], buf.h_new[:2 * n]

    def writeback(self, src, dst, ts, eids, rows, left_row, new_row):
        m, lib, ptr, check = self.model, self.lib, self.ptr, self.check
        dev = m.device
        Bg = len(src)
        s, d_, e = (self._dev(x, torch.int64) for x in (src, dst, eids))
        t32 = self._dev(np.asarray(ts, dtype=np.float64), torch.float64).float()
        pos, ts2 = torch.cat([s, d_]), t32.repeat(2)
        upos, index = self.hip_ops.select_latest_nids(pos, ts2, m.n_nodes)
        n = len(upos)
        n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        rows_new = self._dev(new_row, torch.int64)[index].contiguous()
        rows_left = self._dev(left_row, torch.int64)[index].contiguous()
        ms = m.model_struct()
        st = self.hip_ops.stream_ptr(dev)
        rows = rows.contiguous()
        check(lib.tg_consume_update_right_rows(C.byref(ms), ptr(upos), ptr(n_dev), n, ptr(rows), ptr(rows_new),
                                               ptr(self.err), st), 'tg_consume_update_right_rows')
        check(lib.tg_store_events(C.byref(ms), Bg, ptr(s), ptr(d_), ptr(t32), ptr(e), ptr(upos), ptr(index),
                                  ptr(n_dev), ptr(self.err), st), 'tg_store_events')
        L = m.left_memory
        check(lib.tg_memory_scatter2(n, None, ptr(upos), ptr(rows_left), ptr(index), m.memory_dim, ptr(rows),
                                     ptr(ts2), ptr(L.vals), ptr(L.update_ts), ptr(L.active_mask), 1, ptr(self.err),
                                     st), 'tg_memory_scatter2')

    def check_invariants(self):
        self.hip_ops.raise_if_err(self.err)
        self.hip_ops.raise_if_err(self.buf.err)


# --------------------------------------------------------------------------- benchmark leg
def bench_main(args, cfg, make_stream, build_models, rank, local_rank, world):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: weak scaling,
    global batch = N * B events per step; value = events of all ranks / max-over-ranks time."""
    assert world == args.gpus, f'launch with torchrun: WORLD_SIZE={world} but --gpus {args.gpus}'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    tdist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    B, K, d = cfg['B'], cfg['K'], cfg['d']
    Bg = B * world
    n_steps = args.warmup + args.steps
    E = max(cfg['E'], (n_steps + 2) * Bg)
    stream = make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=d)  # identical on every rank
    model, _ = build_models(stream, d, K, cfg['msg_src'], cfg['upd_src'], restarter='static', device=str(dev))
    owner = balanced_owner_table(stream['n_nodes'], stream['dst'], world)
    cap = int(B * 1.5) + 64
    runner = ShardedRunner(HipBackend(model, cap), owner, rank, world, cap)
    keys = ('src', 'dst', 'neg', 'ts', 'eids')
    batches = [[stream[k][b * Bg:(b + 1) * Bg] for k in keys] for b in range(n_steps)]
    plans = [runner.plan(b[1]) for b in batches]  # input preparation, outside the timed region
    max_shard = max(int(p.counts.max()) for p in plans)
    for b in range(args.warmup):
        runner.step(*batches[b], plan=plans[b])
    torch.cuda.synchronize()
    tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(args.warmup, n_steps):
        runner.step(*batches[b], plan=plans[b])
    torch.cuda.synchronize()
    tdist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    tdist.all_reduce(dt, op=tdist.ReduceOp.MAX)
    runner.backend.check_invariants()
    dt = float(dt.item())
    if rank == 0:
        out = dict(metric='processed interaction-events/sec (memory+aggregate+embed), Wikipedia d=172',
                   value=args.steps * Bg / dt, unit='events/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype='f32', data='synthetic',
                   config=dict(workload=cfg['name'], batch_per_gpu=B, global_batch=Bg, dim=d, n_neighbors=K,
                               msg_src=cfg['msg_src'], upd_src=cfg['upd_src'], n_nodes=stream['n_nodes'], events=E,
                               mode='stream (no_grad) STEP 1-6',
                               parallelism=f'dst-owner event shards x{world}, replicated state, 1 RCCL all-gather/batch',
                               max_shard_events=max_shard),
                   roofline=None, cpu_baseline=None)
        print(json.dumps(out))
    tdist.destroy_process_group()
