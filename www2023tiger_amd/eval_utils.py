"""Mirror of tiger/eval_utils.py for the link-prediction path: `eval_edge_prediction`
(eval_utils.py:15-68) and `warmup` (eval_utils.py:102-129).  The model forward is the same
device path as in training; scores stay on the GPU until the end, where AP / AUC per window of
`mean_over_n_samples` events come from one kernel (`tg_ap_auc`) instead of sklearn round trips.
Node classification and trajectory encoding (eval_utils.py:71-99,132-183) are downstream tasks
outside the scope table."""
import math
import os
import warnings
from typing import Optional

import numpy as np
import torch

from ._lib import check, lib, ptr
from .hip_ops import stream_ptr
from .utils import BackgroundThreadGenerator


def ap_auc_windows(pos_pred: torch.Tensor, neg_pred: torch.Tensor, window: int = 200):
    """sklearn average_precision_score / roc_auc_score of every window of `window` events."""
    n = pos_pred.numel()
    dev = pos_pred.device
    nw = math.ceil(n / window) if n else 0
    ap = torch.zeros(nw, dtype=torch.float64, device=dev)
    auc = torch.zeros(nw, dtype=torch.float64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    pos_pred, neg_pred = pos_pred.float().contiguous(), neg_pred.float().contiguous()
    check(lib.tg_ap_auc(n, window, ptr(pos_pred), ptr(neg_pred), ptr(ap), ptr(auc), ptr(bad), stream_ptr(dev)),
          'tg_ap_auc')
    return ap, auc, bad


def _lazy_restart(model, comp_graph, ts, uptodate_nodes: set, device):
    """eval_utils.py:37-42: restart the involved nodes that are not up to date yet."""
    involved_nodes = comp_graph.np_computation_graph_nodes
    restart_nodes = set(involved_nodes.tolist()) - uptodate_nodes
    r_nids = torch.tensor(sorted(restart_nodes), dtype=torch.long, device=device)
    model.restart(r_nids, torch.full((len(r_nids),), ts.min().item(), device=device))
    uptodate_nodes.update(restart_nodes)


def _resident_plan(model, dl, restart_mode: bool):
    """Can the loop below run as a resident stream?  It can when `dl` is this package's BatchLoader over an
    InteractionData (batches are contiguous index ranges in order, negatives a deterministic stream), the batches
    would take the one-call evaluation step anyway (TIGE._fused_eval_ok, a collator graph whose strategy the step
    samples itself).  The lazy restart of eval_utils.py:37-42 runs with its bookkeeping on the device (below).
    TG_EVAL_RESIDENT=0 switches the form off (the per-batch loop is what the reference's harness does literally)."""
    from .data.data_loader import BatchLoader, GraphCollator, InteractionData
    if os.environ.get('TG_EVAL_RESIDENT', '1') == '0':
        return None
    if restart_mode and (getattr(model, 'restarter_fn', None) is None or os.environ.get('TG_EVAL_RESIDENT_RESTART', '1') == '0'):
        return None
    if type(dl) is not BatchLoader or not isinstance(dl.dataset, InteractionData) or type(dl.collate_fn) is not GraphCollator:
        return None  # any other iterable takes the per-batch loop, as in the reference
    graph = dl.collate_fn.graph
    if getattr(graph, 'strategy', None) not in ('recent_edges', 'recent_nodes') or graph.device.type != 'cuda':
        return None
    if not hasattr(model, '_fused_eval_ok') or not model._fused_eval_ok(graph) or model.device != graph.device:
        return None
    if dl.collate_fn.n_neighbors != model.n_neighbors or dl.collate_fn.n_layers != model.n_layers:
        return None
    lo, hi = dl._range()
    return (lo, hi, graph) if hi > lo and dl.batch_size > 0 else None


def _eval_resident(model, dl, plan, mean_over_n_samples: int, restart_mode: bool = False, uptodate_nodes: Optional[set] = None):
    """eval_edge_prediction's loop (eval_utils.py:29-57) over a RESIDENT stream: the loader's event columns and
    negatives are uploaded once, every batch is one `tg_train_step` call without gradient buffers - the very call
    `contrast_learning` makes per batch under no_grad - that reads its rows at a device-side offset and leaves its
    scores in place in the [N] score columns; nothing else runs on the host per batch (no collation objects, no
    per-column transfers, no per-batch clones / sigmoid / invariant read-back).  With the model's own forms
    (TG_EVAL_STREAM=0) these are the same kernels on the same inputs in the same order as the per-batch loop: scores
    equal bit for bit; by default the pass also streams with eager updates and pre-multiplied weights (below), equal
    to float32 rounding (tests/test_hip_eval.py: test_resident_eval_equals_the_per_batch_loop)."""
    from .model.training import TrainBuffers
    lo, hi, graph = plan
    ds, bs, dev = dl.dataset, dl.batch_size, model.device
    N = hi - lo
    # Parameters are fixed for the whole pass: stream it as INTEGRATION.md's inference recipe does - eager updates and
    # pre-multiplied attention weights (TIGE.eager_updates / fuse_attention: same results as the lazy, unfused forms to
    # float32 rounding) - unless the per-node tables they bring would be large (TG_EVAL_TABLE_BYTES, default 8 GiB:
    # pending rows, query rows, centre rows = n_nodes (2 d + n_head (2 d + d_e)) floats); both switches are put back.
    had = (model._pending is not None, model._fused is not None)
    d, de = model.memory_dim, (model.efeat_dim if model.raw_feat_getter.efeats is not None else 0)
    need = 4 * model.msg_store.n * (2 * d + model.n_head * (2 * d + de))
    budget = int(os.environ.get('TG_EVAL_TABLE_BYTES', str(8 << 30)))
    stream_form = all(had) or (os.environ.get('TG_EVAL_STREAM', '1') != '0' and need <= budget
                               and need <= torch.cuda.mem_get_info(dev)[0] // 2)
    try:
        if stream_form and not all(had):
            if not had[0]:
                model.eager_updates(True)
            if not had[1]:
                model.fuse_attention(True)
        return _eval_resident_run(model, ds, bs, dev, N, lo, hi, graph, TrainBuffers, lean=stream_form,
                                  restart_mode=restart_mode, uptodate_nodes=uptodate_nodes)
    finally:
        if stream_form and not all(had):
            if not had[0]:
                model.eager_updates(False)
            if not had[1]:
                model.fuse_attention(False)


def _restart_listed(model, tb, graph):
    """The lazy restart of one batch (eval_utils.py:37-42) with the bookkeeping on the device: a collate-only pass over
    the batch the step is about to read flags its involved nodes, lists `involved & ~uptodate`, marks them up to date and
    leaves the batch's earliest time (tiger_hip.h: tg_lazy_restart, list form); the host reads back ONE count and hands
    the device-resident list to TIGER.restart - no node set crosses the bus, no Python set arithmetic per batch."""
    import ctypes as C
    cb = tb.sb._lazy_collate
    m = model.model_struct()
    model.prepare_pass(cb, graph)
    check(lib.tg_stream_step(C.byref(m), C.byref(graph.tcsr), C.byref(cb.io), ptr(cb.ws), cb.ws.numel(), stream_ptr(model.device)),
          'tg_stream_step(lazy restart list)')
    n = int(cb.counts[3].item())
    if n:
        nids = tb.sb.lazy_list[:n]
        # a model that streams with eager updates: were its per-node tables current?  Then they follow the restart - the
        # restarted nodes have no pending message any more and new memories: their centre / query rows are recomputed -
        # instead of being rebuilt for every node at the next step
        current = (model._pending is not None and model._pending_stamp == model._state_stamp()
                   and (getattr(model, '_gtab', None) is None or getattr(model, '_gtab_stamp', None) is not None))
        model.restart_list(nids, tb.sb.lazy_tmin)  # (one library call for the SeqRestarter, TIGER.restart otherwise)
        if current:
            model._tables_follow_restart(nids)
    tb.sb.lazy_batch += 1
    return n


class _RestartPipeline:
    """The list-form lazy restart (see _restart_listed) with the collate-only passes run ONE BATCH AHEAD of the steps.
    A pass reads the graph, the batch arrays and the up-to-date bitmap only - nothing a step or a restart writes - so pass
    k + 1 is enqueued before restart k and step k; its count travels to pinned host memory behind it.  When the host
    needs count k + 1 the device has long passed that point of the stream: the read-back no longer drains the queue (it
    did: per batch the device then idled for as long as the host took to enqueue the restarter's dozen launches).
    Two contexts alternate (list, earliest time, count of a pass live until its restart has been enqueued); same lists,
    same order of marks, same results as _restart_listed."""

    def __init__(self, model, tb, graph, first, count):
        self.model, self.tb, self.graph, self.count = model, tb, graph, count
        sb = tb.sb
        self.ctx = [sb._lazy_collate, sb.lazy_collate_context(model)]
        self.offsets = first + torch.arange(count, dtype=torch.int64, device=model.device) * sb.B
        self.host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.ev = [torch.cuda.Event() for _ in range(2)]
        # TG_EVAL_RESTART_GRAPH=1 (off by default): batches whose count fits the capacity replay ONE captured graph per context
        # instead of the eager library calls - the restart with the live count on the device (TIGE.restart_list_captured) -
        # once an eager restart has run (first-use initialisations stay out of a capture).  Measured (C2 shapes, bs 200): no
        # gain - 0.256 against 0.252 ms per batch over 500 batches, and two captures cost a 100-batch pass 5 ms: the pass is
        # bound by the DEVICE time of ~38 short kernels per batch (0.26 ms summed), not by their launches.
        from .model.restarters import SeqRestarter
        self.graph_cap = 256 if (os.environ.get('TG_EVAL_RESTART_GRAPH', '0') != '0'
                                 and isinstance(model.restarter_fn, SeqRestarter) and not model.restarter_fn.training
                                 and model.restarter_fn.graph.strategy == 'recent_edges') else 0
        self.graphs = [None, None]
        self.eager_done = False
        # Two streams (TG_EVAL_RESTART_OVERLAP=0: off).  The restarter's forward of a list reads the graph, the feature tables
        # and its own parameters - nothing a step writes; a pass reads the batch arrays, the graph and the up-to-date bitmap
        # (keep_msg_bits: it leaves the has-message bits to tg_restart_apply).  Both run on a side stream: pass k + 1 and
        # forward k beside step k - 1; the main stream keeps the state: apply k, the table rows of the restarted nodes, step k.
        # Events order the two: the main stream waits for forward k, the side stream for apply k - 1 before pass k + 1
        # overwrites that context's list (and forward k + 1 its rows).  Same lists, same rows, same state as one stream.
        self.overlap = (os.environ.get('TG_EVAL_RESTART_OVERLAP', '1') != '0' and not self.graph_cap
                        and model.restart_list_split_ok() and not bool(sb.lazy_trigger.any()))
        if self.overlap:
            dev, d = model.device, model.memory_dim
            self.main = torch.cuda.current_stream(dev)
            self.side = torch.cuda.Stream(device=dev)
            cap = max(int(cb.lazy_list.numel()) for cb in self.ctx)
            self.rows = [(torch.empty(cap, d, device=dev), torch.empty(cap, d, device=dev), torch.empty(cap, device=dev))
                         for _ in range(2)]
            self.fwd_done = [torch.cuda.Event() for _ in range(2)]
            self.applied = [torch.cuda.Event() for _ in range(2)]
            for cb in self.ctx:
                cb._lazy.keep_msg_bits = 1
            self.side.wait_stream(self.main)
            with torch.cuda.stream(self.side):
                self._pass(0)
        else:
            self._pass(0)

    def _pass(self, k):
        import ctypes as C
        cb = self.ctx[k % 2]
        cb.io.offset_dev = self.offsets.data_ptr() + 8 * k
        m = self.model.model_struct()
        self.model.prepare_pass(cb, self.graph)
        check(lib.tg_stream_step(C.byref(m), C.byref(self.graph.tcsr), C.byref(cb.io), ptr(cb.ws), cb.ws.numel(),
                                 stream_ptr(self.model.device)), 'tg_stream_step(lazy restart list)')
        self.host[k % 2].copy_(cb.counts[3:4], non_blocking=True)
        self.ev[k % 2].record()
        self.tb.sb.lazy_batch += 1

    def close(self):
        """The passes have marked the up-to-date bitmap on the side stream: whoever reads it next is ordered behind them."""
        if self.overlap:
            self.main.wait_stream(self.side)
            for cb in self.ctx:
                cb._lazy.keep_msg_bits = 0

    def _restart_overlapped(self, k):
        model, j = self.model, k % 2
        self.ev[j].synchronize()
        n = int(self.host[j][0])
        cb = self.ctx[j]
        hl, hr, pt = self.rows[j]
        with torch.cuda.stream(self.side):
            if n:
                model.restart_list_forward(cb.lazy_list[:n], cb.lazy_tmin, hl, hr, pt)
                self.fwd_done[j].record()
            if k + 1 < self.count:
                if k >= 1:
                    self.side.wait_event(self.applied[1 - j])  # apply k - 1 and its table rows have read that context's list
                self._pass(k + 1)
        if n:
            self.main.wait_event(self.fwd_done[j])
            current = (model._pending is not None and model._pending_stamp == model._state_stamp()
                       and (getattr(model, '_gtab', None) is None or getattr(model, '_gtab_stamp', None) is not None))
            model.restart_list_apply(cb.lazy_list[:n], hl, hr, pt)
            if current:
                model._tables_follow_restart(cb.lazy_list[:n])
        self.applied[j].record()
        return n

    def restart(self, k):
        """Before step k: enqueue pass k + 1, then the restart of batch k's list."""
        if self.overlap:
            return self._restart_overlapped(k)
        model = self.model
        if k + 1 < self.count:
            self._pass(k + 1)
        self.ev[k % 2].synchronize()
        n = int(self.host[k % 2][0])
        if n:
            cb = self.ctx[k % 2]
            current = (model._pending is not None and model._pending_stamp == model._state_stamp()
                       and (getattr(model, '_gtab', None) is None or getattr(model, '_gtab_stamp', None) is not None))
            cap = min(self.graph_cap, cb.lazy_list.numel())
            if cap and n <= cap and self.eager_done:
                g = self.graphs[k % 2]
                if g is None:
                    g = self.graphs[k % 2] = self._capture(cb, cap)
                model._touch()
                g.replay()
                if current:  # what _tables_follow_restart records on the host
                    if getattr(model, '_gtab', None) is not None:
                        model._gtab_stamp = (model._state_stamp(), tuple(model._attn_stamp()), id(model._fused))
                    model._pending_stamp = model._state_stamp()
            else:
                model.restart_list(cb.lazy_list[:n], cb.lazy_tmin)
                if current:
                    model._tables_follow_restart(cb.lazy_list[:n])
                self.eager_done = True
        return n

    def _capture(self, cb, cap):
        model = self.model
        _ = model.model_struct(), model.restarter_fn._struct()
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=model.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
            model.restart_list_captured(cb.lazy_list[:cap], cb.counts[3:4], cb.lazy_tmin)
        torch.cuda.synchronize()
        return g


class _RestartRun:
    """Batches [k0, count) of the restart-mode pass as ONE library call (tg_eval_restart_run; contract in tiger_hip.h): the
    calls of the per-batch loop - pass, restarter's forward, apply, table rows, step - sequenced by the library on two
    streams, the forward / apply / table rows once per GROUP of batches (their lists are disjoint and a node listed for batch
    k + 1 is not involved in batch k).  The library spends ~3 us of host time per launch where this module spent 0.24 ms per
    batch, and the restarter's dozen latency-bound launches are shared by a group.  SeqRestarter in inference form, no
    pre-drawn triggers.  Memory: two row sets for a group's lists at their bound (the group shrinks until they fit ROWS_LIMIT)
    and the restarter's workspace for FWD_NODES nodes per forward (a group with more takes several forwards)."""
    WS_LIMIT = 2 << 30   # the restarter's workspace (sized for FWD_NODES nodes per forward, halved until it fits)
    ROWS_LIMIT = 2 << 30  # the two row sets of a group (ids, both memories' rows, times)
    FWD_NODES = 2048      # far below the lists' bound (3 B (K + 1) per batch), which only the first batches of a stream approach

    def __init__(self, model, tb, graph, first, count, plan):
        self.model, self.tb, self.graph, self.first, self.count = model, tb, graph, first, count
        self.group, self.fwd_nodes = plan
        self.counts = []

    @staticmethod
    def wanted(model, count):
        """the cheap part of `plan`: a restarter whose forward reads no state (SeqRestarter in inference form over a
        recent-edges graph, StaticRestarter), one layer, state addressed by node id, more than one batch"""
        from .model.restarters import StaticRestarter
        r = model.restarter_fn
        if os.environ.get('TG_EVAL_RESTART_RUN', '1') == '0' or count < 2 or model.n_layers != 1 or model.device.type != 'cuda':
            return False
        if isinstance(r, StaticRestarter):
            # TG_EVAL_RESTART_RUN=2 only: measured equal to the loop inside the step (bs 200: 0.118 against 0.115 ms per batch
            # of the whole harness call), which needs neither contexts nor row sets - that one stays the default
            return os.environ.get('TG_EVAL_RESTART_RUN', '1') == '2' and getattr(model, '_row_of', None) is None
        return model.restart_list_split_ok()

    @staticmethod
    def plan(model, tb, count):
        """-> (batches per group, nodes per forward), or None: the run does not apply"""
        if not _RestartRun.wanted(model, count) or bool(tb.sb.lazy_trigger.any()):
            return None
        import ctypes as C
        from .model.restarters import StaticRestarter
        static = isinstance(model.restarter_fn, StaticRestarter)
        cap, d = int(tb.sb._lazy_collate.lazy_list.numel()), model.memory_dim
        G = max(1, min(8, int(os.environ.get('TG_EVAL_RESTART_GROUP', '8'))))
        while G > 1 and 2 * min(G * cap, model.n_nodes) * (8 * d + 12) > _RestartRun.ROWS_LIMIT:
            G //= 2
        if 2 * min(G * cap, model.n_nodes) * (8 * d + 12) > _RestartRun.ROWS_LIMIT:
            return None
        nodes = min(_RestartRun.FWD_NODES, min(G * cap, model.n_nodes))
        if static:  # (one gather launch per forward, no workspace)
            return G, min(G * cap, model.n_nodes)
        m, rs = model.model_struct(), model.restarter_fn._struct()
        while nodes >= 64:
            if 0 < int(lib.tg_restart_seq_list_workspace_bytes(C.byref(m), C.byref(rs), nodes)) <= _RestartRun.WS_LIMIT:
                return G, nodes
            nodes //= 2
        return None

    def run(self, k0, pos_ptr, neg_ptr):
        """Batches k0 .. count - 1; `pos_ptr` / `neg_ptr`: where batch k0's logits go (those of the later ones behind them)."""
        import ctypes as C
        from ._lib import TgRestartRun
        model, tb, sb, G = self.model, self.tb, self.tb.sb, self.group
        dev, d, nb = model.device, model.memory_dim, self.count - k0
        ctx = [sb._lazy_collate] + [sb.lazy_collate_context(model) for _ in range(2 * G - 1)]
        cap = int(ctx[0].lazy_list.numel())
        rows_cap = min(G * cap, model.n_nodes)
        offsets = self.first + torch.arange(k0, self.count, dtype=torch.int64, device=dev) * sb.B
        host = torch.zeros(2 * G, 16, dtype=torch.int32).pin_memory()
        rows = [(torch.empty(rows_cap, dtype=torch.int64, device=dev), torch.empty(rows_cap, d, device=dev),
                 torch.empty(rows_cap, d, device=dev), torch.empty(rows_cap, device=dev)) for _ in range(2)]
        eager = model._pending is not None
        if eager:  # the protocol of TrainBuffers.launch: tables synchronised before, kept current by the steps themselves
            if model._fused is not None and model._fused_stamp != model._attn_stamp():
                model.fuse_attention()
            model._sync_pending()
            model._sync_gtab()
        from .model.restarters import StaticRestarter
        r = model.restarter_fn
        static = isinstance(r, StaticRestarter)
        m = model.model_struct()
        if static:
            rs, fwd_ws = None, None
        else:
            rs = r._struct()
            nbytes = int(lib.tg_restart_seq_list_workspace_bytes(C.byref(m), C.byref(rs), self.fwd_nodes))
            fwd_ws = torch.empty(nbytes + 1024, dtype=torch.uint8, device=dev)
        gtab_ws = (model._ws('gtab_r', rows_cap * (4 * d + 4) + 64)
                   if (eager and getattr(model, '_gtab', None) is not None) else None)
        n_restarted = np.zeros(nb, dtype=np.int32)
        run = TgRestartRun()
        run.group = G
        for j, cb in enumerate(ctx):
            cb._lazy.keep_msg_bits = 1
            model.prepare_pass(cb, self.graph)
            run.pass_io[j], run.pass_ws[j], run.pass_ws_bytes[j] = C.addressof(cb.io), ptr(cb.ws), cb.ws.numel()
            run.count_host[j] = host[j].data_ptr()
        for j in range(2):
            run.ids[j], run.h_left[j], run.h_right[j], run.prev_ts[j] = (ptr(t) for t in rows[j])
        run.g_restart = C.addressof(model.restarter_fn.graph.tcsr)
        run.offsets, run.batch_dev, run.cap, run.rows_cap = ptr(offsets), ptr(sb.lazy_batch), cap, rows_cap
        run.fwd_nodes = self.fwd_nodes
        if static:
            run.static_left, run.static_right = ptr(r.left_emb.weight), ptr(r.right_emb.weight)
        else:
            run.fwd_ws, run.fwd_ws_bytes = ptr(fwd_ws), fwd_ws.numel()
        if gtab_ws is not None:
            run.gtab_ws, run.gtab_ws_bytes = ptr(gtab_ws), gtab_ws.numel()
        run.pos_scores, run.neg_scores = pos_ptr, neg_ptr
        run.n_restarted = n_restarted.ctypes.data
        if int(tb.sb.io.lean) and os.environ.get('TG_EVAL_PREFETCH', '1') != '0':  # collate prefetch inside the groups (tg_restart_run)
            run.stream_len, run.first_offset = int(sb.src.numel()), int(self.first + k0 * sb.B)
        tb.io.step.rows_hint = model.rows_bound()
        try:
            check(lib.tg_eval_restart_run(C.byref(m), C.byref(self.graph.tcsr), None if static else C.addressof(rs),
                                          C.addressof(tb.io), ptr(tb.ws),
                                          tb.ws.numel(), C.byref(run), nb, stream_ptr(dev)), 'tg_eval_restart_run')
        finally:
            for cb in ctx:
                cb._lazy.keep_msg_bits = 0
        self.counts = n_restarted.tolist()
        model._step_serial = getattr(model, '_step_serial', 0) + nb
        if not eager or any(self.counts):
            model._touch()  # state changed outside the eager streaming step (restarts; every step of a model without tables)
        if eager:  # what _tables_follow_restart records: the restarted nodes' rows were recomputed, the tables are current
            if getattr(model, '_gtab', None) is not None:
                model._gtab_stamp = (model._state_stamp(), tuple(model._attn_stamp()), id(model._fused))
            model._pending_stamp = model._state_stamp()
        self._keep = (ctx, offsets, host, rows, fwd_ws, run)  # (alive until the caller's read-back has drained the stream)


def _eval_resident_run(model, ds, bs, dev, N, lo, hi, graph, TrainBuffers, lean, restart_mode=False, uptodate_nodes=None):
    c = getattr(ds, '_dev', None)
    if c is not None and c['device'] == dev:
        src, dst, ts64, eids = (c[k][lo:hi] for k in ('src', 'dst', 'ts', 'eids'))
        neg = c['neg'][lo:hi] if ds.eval else None
    else:
        i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi], dtype=np.int64)).to(dev)
        src, dst, eids = i64(ds.src), i64(ds.dst), i64(ds.eids)
        ts64 = torch.from_numpy(np.ascontiguousarray(ds.ts[lo:hi], dtype=np.float64)).to(dev)
        neg = i64(ds.neg_dst) if ds.eval else None
    if neg is None:  # a training split: the N draws the per-event __getitem__ calls would make, on the device
        neg = ds.neg_dst_sampler.sample_pairs_device(N, dev)[1]
    resident = tuple(t.contiguous() for t in (src, dst, neg, ts64, eids))
    pos_all = torch.empty(N, dtype=torch.float32, device=dev)
    neg_all = torch.empty(N, dtype=torch.float32, device=dev)
    n_full, rem = divmod(N, bs)
    bufs, runs = [], []
    for B, first, count in ((bs, 0, n_full), (rem, n_full * bs, 1 if rem else 0)):
        if not count:
            continue
        tb = TrainBuffers(model, B, resident=resident, eval_only=True, want_prev=not lean, lean=lean,
                          prefetch=(lean and count > 1 and not restart_mode  # (a restart changes state behind a prefetched collate)
                                    and os.environ.get('TG_EVAL_PREFETCH', '1') != '0'))
        tb.sb.offset.fill_(first)
        if restart_mode:
            # restarting from the first batch on, nobody triggers (eval_utils.py:37-42: every batch restarts what is involved
            # and not yet up to date); the up-to-date set starts as the caller's and is handed from buffer to buffer
            # The static restarter's loop runs INSIDE the evaluation step when that step takes the lean table-backed form (no
            # host round trip at all: tg_lazy_restart, static form) and the step's graph is the restarter's own (the in-step
            # loop looks the previous event time up in the graph the step samples from); every other case: the list form
            from .model.restarters import StaticRestarter
            r = model.restarter_fn
            # (the list form as one library call on two streams - _RestartRun - where it applies: faster than the in-step loop)
            in_step = (lean and isinstance(r, StaticRestarter) and getattr(r, 'graph', None) is graph
                       and os.environ.get('TG_EVAL_RESTART_INSTEP', '1') != '0'
                       and not (dev.type == 'cuda' and os.environ.get('TG_EVAL_RESTART_PIPELINE', '1') != '0'
                                and _RestartRun.wanted(model, count)))
            tb.sb.enable_lazy_restart(model, np.zeros(count, dtype=np.uint8), force_list=not in_step)
            if in_step:
                tb.refresh()  # (the step's io is a copy of the buffer's: it now carries the lazy-restart block)
            tb._restart_in_step = in_step
            tb.sb.lazy_restarting.fill_(1)
            if bufs:
                tb.sb.lazy_uptodate.copy_(bufs[-1].sb.lazy_uptodate)
            elif uptodate_nodes:
                from . import hip_ops
                hip_ops.bitmap_mark(torch.tensor(sorted(uptodate_nodes), dtype=torch.int64, device=dev), tb.sb.lazy_uptodate,
                                    model.n_nodes)
        tb.err_host = None  # one read-back at the end (below)
        bufs.append(tb)
        p0, n0 = pos_all.data_ptr() + 4 * first, neg_all.data_ptr() + 4 * first
        # (replaying captured hipGraphs of several steps was tried here: the pass is bound by the device's dependent launches -
        # bs 200: 72 us per batch eager, 75 us as 16-step graphs, and a capture costs ~10 ms - so the steps are launched eagerly)
        pipe = runner = None
        if (restart_mode and not tb._restart_in_step and dev.type == 'cuda'
                and os.environ.get('TG_EVAL_RESTART_PIPELINE', '1') != '0'):
            plan = _RestartRun.plan(model, tb, count)
            if plan:
                runner = _RestartRun(model, tb, graph, first, count, plan)
                runs.append(runner)
            else:
                pipe = _RestartPipeline(model, tb, graph, first, count)
        for k in range(count):
            tb.io.pos_scores, tb.io.neg_scores = p0 + 4 * k * B, n0 + 4 * k * B
            if runner is not None:
                # batch 0 by the calls of the per-batch loop (its counts size the later steps' launches), the others as one call
                _restart_listed(model, tb, graph)
                tb.launch(graph=graph)
                cnt = tb.sb.counts.tolist()
                model.note_rows(cnt[1], cnt[2])
                runner.run(1, p0 + 4 * B, n0 + 4 * B)
                break
            if pipe is not None:
                pipe.restart(k)
            elif restart_mode and not tb._restart_in_step:
                _restart_listed(model, tb, graph)
            tb.launch(graph=graph)
            if k == 0 and count > 8:  # one early read-back: the updater's launches are sized by the counts seen so far
                cnt = tb.sb.counts.tolist()
                model.note_rows(cnt[1], cnt[2])
        if pipe is not None:
            pipe.close()
    words = [(int(tb.sb.err.item()), tb.sb.counts.tolist()) for tb in bufs]  # (also drains the stream)
    for word, cnt in words:
        model.note_rows(cnt[1], cnt[2])
        if word:
            from ._lib import raise_invariants
            raise_invariants(word & 0xFFFFFFFF)
    if restart_mode and uptodate_nodes is not None and bufs:  # updated in place, as in the reference
        from . import hip_ops
        comp = hip_ops.unique_compact(bufs[-1].sb.lazy_uptodate, model.n_nodes, model.n_nodes)
        uptodate_nodes.update(comp['ids'][:int(comp['count'].item())].tolist())
    return pos_all.sigmoid_(), neg_all.sigmoid_()


def eval_edge_prediction(model, dl, device: torch.device, restart_mode: bool, uptodate_nodes: Optional[set] = None,
                         mean_over_n_samples: int = 200):
    """-> (mean AP, mean AUC) over windows of `mean_over_n_samples` events.  `uptodate_nodes` is
    updated in place when given (as in the reference)."""
    model.eval()
    uptodate_nodes = set() if uptodate_nodes is None else uptodate_nodes
    plan = _resident_plan(model, dl, restart_mode)
    if plan is not None:
        with torch.no_grad():
            model._poll_train_errors()
            pos_pred, neg_pred = _eval_resident(model, dl, plan, mean_over_n_samples, restart_mode, uptodate_nodes)
        ap, auc, bad = ap_auc_windows(pos_pred, neg_pred, mean_over_n_samples)
        if int(bad.item()):
            warnings.warn(f'Encounter invalid values: {int(bad.item())} non-finite predictions were dropped')
        return float(ap.mean().item()), float(auc.mean().item())
    pos_all, neg_all = [], []
    with torch.no_grad():
        for src_ids, dst_ids, neg_dst_ids, ts, eids, _, comp_graph in BackgroundThreadGenerator(dl):
            src_ids, dst_ids, neg_dst_ids = (x.long().to(device) for x in (src_ids, dst_ids, neg_dst_ids))
            ts, eids = ts.float().to(device), eids.long().to(device)
            comp_graph.to(device)
            if restart_mode:
                _lazy_restart(model, comp_graph, ts, uptodate_nodes, device)
            _, _, pos_scores, neg_scores, *_ = model.contrast_learning(src_ids, dst_ids, neg_dst_ids, ts, eids,
                                                                       comp_graph)
            pos_all.append(pos_scores.sigmoid())
            neg_all.append(neg_scores.sigmoid())
    model._poll_train_errors()  # the last batch's invariant word (the one-call step reads it back asynchronously)
    if not pos_all:
        return float('nan'), float('nan')
    ap, auc, bad = ap_auc_windows(torch.cat(pos_all), torch.cat(neg_all), mean_over_n_samples)
    if int(bad.item()):
        warnings.warn(f'Encounter invalid values: {int(bad.item())} non-finite predictions were dropped')
    return float(ap.mean().item()), float(auc.mean().item())


def warmup(model, dl, device: torch.device, uptodate_nodes: Optional[set] = None):
    """Only valid in restart mode: stream the loader through the model with lazy restarts."""
    model.eval()
    uptodate_nodes = set() if uptodate_nodes is None else uptodate_nodes
    plan = _resident_plan(model, dl, True)
    if plan is not None:  # the same pass as eval_edge_prediction(restart_mode=True), scores unused
        with torch.no_grad():
            model._poll_train_errors()
            _eval_resident(model, dl, plan, 200, True, uptodate_nodes)
        return uptodate_nodes
    with torch.no_grad():
        for src_ids, dst_ids, neg_dst_ids, ts, eids, _, comp_graph in BackgroundThreadGenerator(dl):
            src_ids, dst_ids, neg_dst_ids = (x.long().to(device) for x in (src_ids, dst_ids, neg_dst_ids))
            ts, eids = ts.float().to(device), eids.long().to(device)
            comp_graph.to(device)
            _lazy_restart(model, comp_graph, ts, uptodate_nodes, device)
            model.contrast_learning(src_ids, dst_ids, neg_dst_ids, ts, eids, comp_graph)
    model._poll_train_errors()
    return uptodate_nodes
