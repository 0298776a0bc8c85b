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
import time
from typing import Optional, Tuple

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
    gathered buffer.  Rank r's send buffer holds `rows_per_rank` rows; the h(t-) row of
    (role, slot) is at left_base + role*role_stride + slot and the h(t'+) row at
    new_base + role*role_stride + slot.

    balance=False: an event runs on owner[dst] (shards are ragged, bounded by `cap`).
    balance=True : every rank gets exactly Bg/world events - an event runs on owner[dst]
                   while that rank has room and spills to the least loaded rank otherwise.
                   State is replicated, so placement only affects load, never results;
                   exact balance gives static shapes (one captured hipGraph per rank)."""

    def __init__(self, dst: np.ndarray, owner: np.ndarray, world: int, cap: int, balance: bool = False,
                 layout: Optional[Tuple[int, int, int, int]] = None):
        Bg = len(dst)
        self.Bg, self.world, self.cap = Bg, world, cap
        pref = owner[dst]
        if balance:
            if Bg % world:
                raise ValueError('balanced plan needs the global batch to divide by the world size')
            per = Bg // world
            if per > cap:
                raise ValueError(f'shard of {per} events exceeds capacity {cap}')
            rank_of = np.empty(Bg, dtype=np.int64)
            room = np.full(world, per, dtype=np.int64)
            spill = []
            for e in range(Bg):  # stream order
                r = pref[e]
                if room[r] > 0:
                    rank_of[e] = r
                    room[r] -= 1
                else:
                    spill.append(e)
            for e in spill:
                r = int(np.argmax(room))  # most room first (ties -> lowest rank)
                rank_of[e] = r
                room[r] -= 1
        else:
            rank_of = pref
        self.rank_of = rank_of
        self.counts = np.bincount(rank_of, minlength=world).astype(np.int64)
        if self.counts.max() > cap:
            raise ValueError(f'shard of {self.counts.max()} events exceeds capacity {cap}')
        order = np.argsort(rank_of, kind='stable')  # events grouped by rank, stream order kept inside a rank
        starts = np.concatenate([[0], np.cumsum(self.counts)[:-1]])
        slot = np.empty(Bg, dtype=np.int64)
        slot[order] = np.arange(Bg) - np.repeat(starts, self.counts)
        self.local_idx = [order[starts[r]:starts[r] + self.counts[r]] for r in range(world)]
        rows_per_rank, left_base, new_base, role_stride = layout or (4 * cap, 0, 2 * cap, cap)
        role = np.repeat(np.array([0, 1]), Bg)
        r2, s2 = np.tile(rank_of, 2), np.tile(slot, 2)
        self.left_row = r2 * rows_per_rank + left_base + role * role_stride + s2  # [2Bg], position i of cat[src,dst]
        self.new_row = r2 * rows_per_rank + new_base + role * role_stride + s2


def all_gather_rows(send: torch.Tensor, world: int, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """send [...] -> [world, ...].  On RCCL the gather runs on device; gloo (CPU tests, or
    two ranks sharing one GPU) stages through host memory."""
    if out is None:
        out = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
    if tdist.get_backend(group) == 'nccl':
        tdist.all_gather_into_tensor(out, send.contiguous(), group=group)
        return out
    parts = [torch.empty(send.shape, dtype=send.dtype) for _ in range(world)]
    tdist.all_gather(parts, send.detach().cpu().contiguous(), group=group)
    out.copy_(torch.stack(parts).to(send.device))
    return out


class ShardedRunner:
    """Drives one global batch: local embed -> all-gather -> replicated write-back.
    `backend` supplies the two compute halves (HipBackend in production; the CPU tests
    plug in an oracle-backed object to exercise this host logic under gloo)."""

    def __init__(self, backend, owner: np.ndarray, rank: int, world: int, cap: int, group=None,
                 balance: bool = False):
        self.backend, self.owner, self.rank, self.world, self.cap, self.group = backend, owner, rank, world, cap, group
        self.balance = balance

    def plan(self, dst: np.ndarray) -> ShardPlan:
        return ShardPlan(np.asarray(dst), self.owner, self.world, self.cap, balance=self.balance)

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
    """The two halves on the HIP engine, driven with host arrays (ragged shards allowed):
    tg_stream_step(embed_only) and tg_stream_writeback."""

    def __init__(self, model, cap: int):
        from . import hip_ops
        from ._lib import TgWritebackIo, check, lib, ptr
        self.model, self.cap = model, cap
        self.hip_ops, self.check, self.lib, self.ptr, self.WbIo = hip_ops, check, lib, ptr, TgWritebackIo
        self.buf = model.StepBuffers(model, cap, False, embed_only=True)
        self.err = hip_ops.new_err(model.device)
        self._wb_ws = None

    def _dev(self, a, dt):
        return torch.as_tensor(a).to(self.model.device, dt).contiguous()

    def embed(self, src, dst, neg, ts, eids):
        n, buf = len(src), self.buf
        if n == 0:
            z = torch.zeros(0, self.model.memory_dim, device=self.model.device)
            return z, z
        buf.src[:n], buf.dst[:n], buf.neg[:n] = (self._dev(x, torch.int64) for x in (src, dst, neg))
        buf.ts[:n], buf.eids[:n] = self._dev(ts, torch.float64), self._dev(eids, torch.int64)
        buf.io.B = n
        self.model.launch_step(buf)
        return buf.h[:2 * n], buf.h_new[:2 * n]

    def writeback(self, src, dst, ts, eids, rows, left_row, new_row):
        m, lib, ptr = self.model, self.lib, self.ptr
        m._touch()  # state changes outside the model's own step
        Bg = len(src)
        ms = m.model_struct()
        keep = [self._dev(src, torch.int64), self._dev(dst, torch.int64), self._dev(ts, torch.float64),
                self._dev(eids, torch.int64), rows.contiguous(), self._dev(left_row, torch.int64),
                self._dev(new_row, torch.int64)]
        nbytes = int(lib.tg_stream_writeback_workspace_bytes(C.byref(ms), Bg))
        if self._wb_ws is None or self._wb_ws.numel() < nbytes:
            self._wb_ws = torch.empty(nbytes, dtype=torch.uint8, device=m.device)
        io = self.WbIo(Bg, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), None, 0, 0, ptr(keep[4]),
                       ptr(keep[5]), ptr(keep[6]), ptr(self.err))
        self.check(lib.tg_stream_writeback(C.byref(ms), C.byref(io), ptr(self._wb_ws), self._wb_ws.numel(),
                                           self.hip_ops.stream_ptr(m.device)), 'tg_stream_writeback')

    def check_invariants(self):
        self.hip_ops.raise_if_err(self.err)
        self.hip_ops.raise_if_err(self.buf.err)


class ResidentShardedStream:
    """Production form for a stream that is resident in HBM: balanced plans are prepared for
    every step up front, each rank's shard of the stream and the global stream sit on the
    device, and a step is  [hipGraph: embed shard] -> all-gather -> [hipGraph: write-back].
    The step writes its outputs straight into one buffer of 5B rows,
    [h(t'+) src | h(t'+) dst | h(t-) src | h(t-) dst | h(neg)]; the first 4B rows are what the other ranks'
    write-back needs and what travels (the negatives' embeddings are outputs of the owner only)."""

    def __init__(self, model, stream: dict, owner: np.ndarray, rank: int, world: int, B: int, n_steps: int,
                 group=None, use_graphs: bool = True):
        from . import hip_ops
        from ._lib import TgWritebackIo, check, lib, ptr
        self.model, self.rank, self.world, self.B, self.group = model, rank, world, B, group
        self.check, self.lib, self.ptr, self.hip_ops = check, lib, ptr, hip_ops
        dev, d = model.device, model.memory_dim
        Bg = B * world
        self.Bg, self.n_steps = Bg, n_steps
        layout = (4 * B, 2 * B, 0, B)  # rows per rank, base of h(t-), base of h(t'+), stride between src and dst rows
        keys = ('src', 'dst', 'neg', 'ts', 'eids')
        local = {k: [] for k in keys}
        left_rows, new_rows = [], []
        for b in range(n_steps):
            sl = slice(b * Bg, (b + 1) * Bg)
            plan = ShardPlan(stream['dst'][sl], owner, world, B, balance=True, layout=layout)
            li = plan.local_idx[rank]
            for k in keys:
                local[k].append(stream[k][sl][li])
            left_rows.append(plan.left_row)
            new_rows.append(plan.new_row)
        tod = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.concatenate(a))).to(dev, dt)
        self.local = tuple(tod(local[k], torch.float64 if k == 'ts' else torch.int64) for k in keys)
        n_ev = n_steps * Bg
        self.glob = tuple(torch.from_numpy(np.ascontiguousarray(stream[k][:n_ev])).to(dev)
                          for k in ('src', 'dst', 'ts', 'eids'))
        self.left_row, self.new_row = tod(left_rows, torch.int64), tod(new_rows, torch.int64)
        self.send = torch.zeros(5 * B, d, dtype=torch.float32, device=dev)
        self.gathered = torch.zeros(world, 4 * B, d, dtype=torch.float32, device=dev)
        self.buf = model.StepBuffers(model, B, False, resident=self.local, embed_only=True, h_out=self.send[2 * B:],
                                     h_new_out=self.send[:2 * B])
        self.err = hip_ops.new_err(dev)
        ms = model.model_struct()
        nbytes = int(lib.tg_stream_writeback_workspace_bytes(C.byref(ms), Bg))
        self.wb_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.wb_off = torch.zeros(1, dtype=torch.int64, device=dev)
        g = self.glob
        self.wb_io = TgWritebackIo(Bg, ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]), ptr(self.wb_off), 1, 0,
                                   ptr(self.gathered), ptr(self.left_row), ptr(self.new_row), ptr(self.err))
        self.g_embed = self.g_wb = None
        self.use_graphs = use_graphs
        self.steps_done = 0
        # graph replays and the collective are ordered through an explicit stream: on the legacy null stream the
        # order between a hipGraph launch and the event the process group records for its own stream is not
        # reliable (a replay around the all-gather faulted when nothing else synchronised the phases)
        self.stream = torch.cuda.Stream(device=dev) if (use_graphs and dev.type == 'cuda') else None

    def _launch_wb(self):
        self.model._touch()  # state changes outside the model's own step
        ms = self.model.model_struct()
        self.check(self.lib.tg_stream_writeback(C.byref(ms), C.byref(self.wb_io), self.ptr(self.wb_ws),
                                                self.wb_ws.numel(), self.hip_ops.stream_ptr(self.model.device)),
                   'tg_stream_writeback')

    def capture(self):
        """Capture both halves into hipGraphs (call after at least one eager step)."""
        side = torch.cuda.Stream(device=self.model.device)
        off_e, off_w = self.buf.offset.clone(), self.wb_off.clone()
        self.g_embed, self.g_wb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: with the default (global) capture mode ANY thread's unsafe runtime call invalidates a
        # capture in progress, and the RCCL process group runs a watchdog thread that polls the events of
        # recent collectives (hipEventQuery) - an intermittently broken capture whose replay then faults
        torch.cuda.synchronize()
        with torch.cuda.graph(self.g_embed, stream=side, capture_error_mode='thread_local'):
            self.model.launch_step(self.buf)
        with torch.cuda.graph(self.g_wb, stream=side, capture_error_mode='thread_local'):
            self._launch_wb()
        self.buf.offset.copy_(off_e)  # capture does not execute, but be explicit
        self.wb_off.copy_(off_w)
        if self.stream is not None:
            torch.cuda.synchronize()  # the eager steps ran on the current stream; replays continue on self.stream

    def step(self, debug: bool = False):
        assert self.steps_done < self.n_steps, 'resident stream exhausted'

        def mark(what):
            if debug:
                torch.cuda.synchronize()
                print(f'[dist debug] step {self.steps_done} {what} ok', flush=True)
        import contextlib
        own = self.stream is not None and self.g_embed is not None
        ctx = torch.cuda.stream(self.stream) if own else contextlib.nullcontext()
        if own:  # whatever the caller queued on its stream happens before this step ...
            self.stream.wait_stream(torch.cuda.current_stream())
        with ctx:
            if self.g_embed is not None:
                self.g_embed.replay()
            else:
                self.model.launch_step(self.buf)
            mark('embed')
            all_gather_rows(self.send[:4 * self.B], self.world, self.group, out=self.gathered)
            mark('all_gather')
            if self.g_wb is not None:
                self.g_wb.replay()
            else:
                self._launch_wb()
            mark('writeback')
        if own:  # ... and whatever it queues next (reading the memories back, say) after it
            torch.cuda.current_stream().wait_stream(self.stream)
        self.steps_done += 1

    def check_invariants(self):
        self.hip_ops.raise_if_err(self.err)
        self.hip_ops.raise_if_err(self.buf.err)


# --------------------------------------------------------------------------- benchmark leg
def bench_main(args, cfg, make_stream, build_models, rank, local_rank, world):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: weak scaling,
    global batch = N * B events per step; value = events of all ranks / max-over-ranks time."""
    assert world == args.gpus, f'launch with torchrun: WORLD_SIZE={world} but --gpus {args.gpus}'
    # RCCL prints a version banner on fd 1 when the communicator is created; the contract is ONE JSON
    # line on stdout, so fd 1 is pointed at stderr until the result is ready
    import os
    import sys
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    # rehearsal on a box with fewer GPUs than ranks (TG_BENCH_REHEARSAL=1): every rank uses GPU 0 and the exchange
    # goes through gloo - the timings mean nothing, the flow (plans, shapes, collectives, the JSON line) is the same
    rehearsal = bool(os.environ.get('TG_BENCH_REHEARSAL'))
    dev = torch.device('cuda', 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if rehearsal:
        tdist.init_process_group('gloo', rank=rank, world_size=world)
    else:
        tdist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    B, K, d = cfg['B'], cfg['K'], cfg['d']
    Bg = B * world
    n_steps = args.warmup + args.steps
    E = max(cfg['E'], (n_steps + 2) * Bg)
    stream = make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=d)  # identical on every rank
    model, _ = build_models(stream, d, K, cfg['msg_src'], cfg['upd_src'], restarter='static', device=str(dev))
    if not getattr(args, 'no_fuse', False):
        model.fuse_attention()  # fixed parameters: pre-multiplied attention weights, as in the 1-GPU bench
    owner = balanced_owner_table(stream['n_nodes'], stream['dst'], world)
    use_graphs = bool(getattr(args, 'dist_graphs', False)) and not args.no_graph
    rs = ResidentShardedStream(model, stream, owner, rank, world, B, n_steps, use_graphs=use_graphs)
    spilled = 0.0
    for b in range(min(n_steps, 50)):  # how often the owner's shard was full (reported, not timed)
        sl = slice(b * Bg, (b + 1) * Bg)
        p = ShardPlan(stream['dst'][sl], owner, world, B, balance=True)
        spilled += float((p.rank_of != owner[stream['dst'][sl]]).mean())
    spilled /= min(n_steps, 50)
    debug = bool(os.environ.get('TG_DIST_DEBUG'))
    n_eager = min(2, args.warmup)
    for _ in range(n_eager):
        rs.step(debug)
    torch.cuda.synchronize()
    if use_graphs:
        rs.capture()
    for _ in range(args.warmup - n_eager):
        rs.step(debug)
    torch.cuda.synchronize()
    tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rs.step()
    torch.cuda.synchronize()
    tdist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    tdist.all_reduce(dt, op=tdist.ReduceOp.MAX)
    rs.check_invariants()
    dt = float(dt.item())
    if rank == 0:
        out = dict(metric='processed interaction-events/sec (memory+aggregate+embed), Wikipedia d=172',
                   value=args.steps * Bg / dt, unit='events/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype='f32', data='synthetic',
                   config=dict(workload=cfg['name'], batch_per_gpu=B, global_batch=Bg, dim=d, n_neighbors=K,
                               msg_src=cfg['msg_src'], upd_src=cfg['upd_src'], n_nodes=stream['n_nodes'], events=E,
                               mode='stream (no_grad) STEP 1-6',
                               parallelism=f'dst-owner event shards x{world} (capacity-balanced), replicated state, '
                                           f'1 RCCL all-gather of {4 * B}x{d} f32 rows per rank per batch',
                               spilled_event_fraction=round(spilled, 4),
                               launch='2 hipGraphs + 1 all-gather per step' if use_graphs else 'eager launches + 1 all-gather per step'),
                   roofline=None, cpu_baseline=None)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    tdist.destroy_process_group()
