"""Mirror of tiger/model/tiger.py: `TIGE` / `TIGER` with the reference's constructor,
method names, buffer/parameter names (state_dict compatible, alias keys included),
executed by libtiger_hip.so.

Ways to run a batch:
  * `contrast_learning(src, dst, neg, ts, eids, computation_graph)` - the reference
    signature.  In train() mode with autograd on, the whole iteration (collate, STEP 1-7,
    backward, write-back; `contrast_and_mutual_learning` adds the mutual loss) is one
    tg_train_step and the returned losses carry an autograd node that hands the finished
    gradients to the parameters.  In eval() / no_grad mode the batch is one tg_train_step
    without gradient buffers; where that call does not apply (non-default sampling strategy,
    a foreign ComputationGraph) STEP 1-6 run as one HIP entry point per reference method
    and STEP 7 in torch.
  * `stream_step(src, dst, neg, ts64, eids)` - collation and STEP 1-6 fused behind one
    C call (tg_stream_step) with no host synchronisation; the benchmarked path.
  * `fuse_attention()` - inference with fixed parameters: pre-multiplied attention weights.
"""
import ctypes as C
from typing import Tuple, Union

import numpy as np
import os

import torch
from torch import Tensor, nn

from .. import hip_ops
from .._lib import TgLazyRestart, TgLinear, TgModel, TgStepIo, check, lib, ptr
from ..hip_ops import stream_ptr
from .basic_modules import MergeLayer
from .memory import Memory, MessageStoreNoGradLastOnly
from .message_modules import (IdentityMessageFunction, LastMessageAggregatorNoGradLastOnly, LinearMessageFunction,
                              MLPMessageFunction)
from .temporal_agg_modules import GraphAttnEmbedding
from .time_encoding import TimeEncode
from .update_modules import GRUUpdater, MergeUpdater
from .utils import select_latest_nids

_TSFM = {'id': 0, 'linear': 1, 'mlp': 2}
_UPD = {'gru': 0, 'merge': 1}


class TIGE(nn.Module):
    _born_rows = None

    @staticmethod
    def born_with_rows(n_rows: int):
        """Context: models constructed inside allocate `n_rows` rows per state table instead of one per node - a rank of the
        physically partitioned multi-GPU layout, whose engine installs the node -> row map right after (dist.py:
        HipPartitionEngine.partition).  Until then such a model must not run."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, TIGE._born_rows = TIGE._born_rows, int(n_rows)
            try:
                yield
            finally:
                TIGE._born_rows = prev
        return ctx()

    def __init__(self, *, raw_feat_getter, graph, n_neighbors: int = 20, n_layers: int = 2, n_head: int = 2,
                 dropout: float = 0.1, msg_src: str, upd_src: str, msg_tsfm_type: str = 'id',
                 mem_update_type: str = 'gru', tgn_mode: bool = True, msg_last_only: bool = True,
                 hit_type: str = 'none'):
        super().__init__()
        if not msg_last_only:
            raise NotImplementedError('only the last-message mailbox is built (init_utils.py:166)')
        self.raw_feat_getter = raw_feat_getter
        self.n_nodes = raw_feat_getter.n_nodes
        self.nfeat_dim = raw_feat_getter.nfeat_dim
        self.efeat_dim = raw_feat_getter.efeat_dim
        self.time_encoder = TimeEncode(dim=self.nfeat_dim)
        self.tfeat_dim = self.time_encoder.dim
        self.memory_dim = self.nfeat_dim
        self.raw_msg_dim = self.memory_dim * 2 + self.efeat_dim + self.tfeat_dim
        self.n_neighbors, self.n_layers, self.n_head = n_neighbors, n_layers, n_head
        self.msg_src, self.upd_src = msg_src, upd_src
        self.tgn_mode, self.msg_last_only = True, True
        self._sanity_check()

        rows = TIGE._born_rows if TIGE._born_rows is not None else self.n_nodes  # (born_with_rows: a partitioned rank)
        self.left_memory = Memory(rows, self.memory_dim)
        self.right_memory = Memory(rows, self.memory_dim)
        self.msg_store = MessageStoreNoGradLastOnly(rows, dim=self.raw_msg_dim)
        # module aliases: they appear as extra state_dict keys exactly as in the reference
        self.msg_memory = self.left_memory if msg_src == 'left' else self.right_memory
        self.upd_memory = self.left_memory if upd_src == 'left' else self.right_memory
        self.msg_aggregate_fn = LastMessageAggregatorNoGradLastOnly(raw_feat_getter=raw_feat_getter,
                                                                   time_encoder=self.time_encoder)
        fn = {'id': IdentityMessageFunction, 'linear': LinearMessageFunction, 'mlp': MLPMessageFunction}
        if msg_tsfm_type not in fn:
            raise NotImplementedError(msg_tsfm_type)
        self.msg_tsfm_type = msg_tsfm_type
        self.msg_transform_fn = fn[msg_tsfm_type](raw_msg_dim=self.raw_msg_dim)
        self.msg_dim = self.msg_transform_fn.output_size
        if mem_update_type == 'gru':
            self.right_mem_updater = GRUUpdater(self.msg_dim, self.memory_dim)
        elif mem_update_type == 'merge':
            self.right_mem_updater = MergeUpdater(self.msg_dim, self.memory_dim)
        else:
            raise NotImplementedError(mem_update_type)
        self.mem_update_type = mem_update_type
        self.temporal_embedding_fn = GraphAttnEmbedding(raw_feat_getter=raw_feat_getter,
                                                        time_encoder=self.time_encoder, graph=graph,
                                                        n_neighbors=n_neighbors, n_layers=n_layers, n_head=n_head,
                                                        dropout=dropout)
        self.hit_type = hit_type
        if hit_type == 'vec':
            merge_dim = self.nfeat_dim + self.n_neighbors
        elif hit_type == 'bin':
            self.hit_embedding = nn.Embedding(2, self.nfeat_dim)
            merge_dim = self.nfeat_dim
        elif hit_type == 'count':
            self.hit_embedding = nn.Embedding(self.n_neighbors + 1, self.nfeat_dim)
            merge_dim = self.nfeat_dim
        else:
            merge_dim = self.nfeat_dim
        self.score_fn = MergeLayer(merge_dim, merge_dim, self.nfeat_dim, 1, dropout=dropout)
        self.contrast_loss_fn = nn.BCEWithLogitsLoss()
        self._struct_cache = None
        self._fused = None
        self._pending = None        # eager updates: table of precomputed updater rows (see eager_updates)
        self._pending_stamp = None  # state stamp the table is current for
        self._state_version = 0     # bumped by every method that changes state outside the eager streaming step
        self._step_ws = {}

    def _sanity_check(self):
        if self.msg_src not in {'left', 'right'}:
            raise ValueError(f'Invalid msg_src={self.msg_src}')
        if self.upd_src not in {'left', 'right'}:
            raise ValueError(f'Invalid upd_src={self.upd_src}')

    # ---- plumbing ---------------------------------------------------------------------
    def restart_list(self, nids: Tensor, t_dev: Tensor):
        """`restart(nids, t.expand(n))` for a device-resident id list and ONE device-resident time, with the SeqRestarter
        (inference form, or train() mode with its dropout), as a single library call (tg_restart_seq_list(_train): histories, anonymised ids, the restarter's
        forward, the state update) - the loops that restart per batch (eval_utils: lazy restart) were bound by the host
        side of the dozen calls this replaces.  Anything else takes `restart`."""
        from .restarters import SeqRestarter
        r = self.restarter_fn
        n = int(nids.numel())
        # (whatever strategy the restarter's graph samples neighbourhoods with, histories are recent-edges lists: graph.py:150-155)
        if (n == 0 or not isinstance(r, SeqRestarter) or getattr(self, '_row_of', None) is not None
                or self.device.type != 'cuda' or t_dev.dtype != torch.float32):
            return self.restart(nids, t_dev.expand(n))
        p = float(r.mha_fn.dropout)
        train = r.training and p > 0
        if train and (r.rng_fn is None or float(r.merger.dropout.p) != p):
            return self.restart(nids, t_dev.expand(n))
        self._touch()
        m, rs = self.model_struct(), r._struct()
        nbytes = int(lib.tg_restart_seq_list_workspace_bytes(C.byref(m), C.byref(rs), n))
        ws = getattr(self, '_restart_list_ws', None)
        if ws is None or ws.numel() < nbytes:
            ws = self._restart_list_ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=self.device)
        if train:  # train() mode: attention / merger dropout is active, as when the reference restarts inside its training loop
            check(lib.tg_restart_seq_list_train(C.byref(m), C.byref(r.graph.tcsr), C.byref(rs), n, ptr(nids), ptr(t_dev), p,
                                                ptr(r.rng_fn()), ptr(ws), ws.numel(), stream_ptr(self.device)),
                  'tg_restart_seq_list_train')
            return
        check(lib.tg_restart_seq_list(C.byref(m), C.byref(r.graph.tcsr), C.byref(rs), n, ptr(nids), ptr(t_dev), ptr(ws),
                                      ws.numel(), stream_ptr(self.device)), 'tg_restart_seq_list')

    def restart_list_split_ok(self) -> bool:
        """Can `restart_list` be taken apart into `restart_list_forward` (reads graph / features / restarter parameters only)
        and `restart_list_apply` (the state update)?  The SeqRestarter in inference form."""
        from .restarters import SeqRestarter
        r = self.restarter_fn
        return (isinstance(r, SeqRestarter) and not (r.training and float(r.mha_fn.dropout) > 0)
                and getattr(self, '_row_of', None) is None and self.device.type == 'cuda')

    def restart_list_forward(self, nids: Tensor, t_dev: Tensor, h_left: Tensor, h_right: Tensor, prev_ts: Tensor, ws_key='a'):
        """The restarter's rows of `restart_list` WITHOUT the state update (tg_restart_seq_list_fwd), on the current stream,
        into the caller's buffers.  Nothing a streaming step writes is read: callable on a second stream beside a step."""
        r = self.restarter_fn
        n = int(nids.numel())
        m, rs = self.model_struct(), r._struct()
        nbytes = int(lib.tg_restart_seq_list_workspace_bytes(C.byref(m), C.byref(rs), n))
        store = self.__dict__.setdefault('_restart_fwd_ws', {})
        ws = store.get(ws_key)
        if ws is None or ws.numel() < nbytes:
            ws = store[ws_key] = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=self.device)
        check(lib.tg_restart_seq_list_fwd(C.byref(m), C.byref(r.graph.tcsr), C.byref(rs), n, ptr(nids), None, ptr(t_dev),
                                          ptr(h_left), ptr(h_right), ptr(prev_ts), ptr(ws), ws.numel(),
                                          stream_ptr(self.device)), 'tg_restart_seq_list_fwd')

    def restart_lists_forward(self, lists, t_devs):
        """`restart_list_forward` over several device-resident lists at once, each at its own device-resident time
        (tg_restart_seq_lists_fwd: one forward; at most 8 lists) -> (ids, h_left, h_right, prev_ts): the lists concatenated
        (empty ones skipped) and their rows.  The lists of consecutive batches of a restart loop are independent of each other."""
        r, dev, d = self.restarter_fn, self.device, self.memory_dim
        counts = [int(x.numel()) for x in lists]
        n = sum(counts)
        ids = torch.empty(n, dtype=torch.int64, device=dev)
        hl, hr, pt = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev), torch.empty(n, device=dev)
        if n == 0:
            return ids, hl, hr, pt
        m, rs = self.model_struct(), r._struct()
        nbytes = int(lib.tg_restart_seq_list_workspace_bytes(C.byref(m), C.byref(rs), n))
        ws = torch.empty(nbytes + 1024, dtype=torch.uint8, device=dev)
        k = len(lists)
        lp = (C.c_void_p * k)(*[ptr(x) for x in lists])
        tp = (C.c_void_p * k)(*[ptr(t) for t in t_devs])
        cn = (C.c_int64 * k)(*counts)
        check(lib.tg_restart_seq_lists_fwd(C.byref(m), C.byref(r.graph.tcsr), C.byref(rs), k, lp, cn, tp, ptr(ids), ptr(hl), ptr(hr),
                                           ptr(pt), ptr(ws), ws.numel(), stream_ptr(dev)), 'tg_restart_seq_lists_fwd')
        return ids, hl, hr, pt

    def restart_list_apply(self, nids: Tensor, h_left: Tensor, h_right: Tensor, prev_ts: Tensor):
        """The state update of `restart_list` (tiger.py:603,608-609) from rows `restart_list_forward` left."""
        self._touch()
        m = self.model_struct()
        check(lib.tg_restart_apply(C.byref(m), int(nids.numel()), ptr(nids), ptr(h_left), ptr(h_right), ptr(prev_ts),
                                   stream_ptr(self.device)), 'tg_restart_apply')

    def restart_list_captured(self, nids_cap: Tensor, n_dev: Tensor, t_dev: Tensor):
        """The device half of `restart_list` + `_tables_follow_restart` with the live count ON THE DEVICE (n_dev, int32): launches
        sized for the capacity len(nids_cap), the first n_dev entries restarted (tg_restart_seq_list_dev, tg_attn_gtab_rows with
        its device count).  No host value depends on the count: callable under stream capture; the caller replays the graph
        for every batch whose count fits and does the host-side bookkeeping itself (eval_utils._RestartPipeline)."""
        r = self.restarter_fn
        cap = int(nids_cap.numel())
        m, rs = self.model_struct(), r._struct()
        nbytes = int(lib.tg_restart_seq_list_workspace_bytes(C.byref(m), C.byref(rs), cap))
        ws = getattr(self, '_restart_cap_ws', None)
        if ws is None or ws.numel() < nbytes:
            ws = self._restart_cap_ws = torch.empty(nbytes + 1024, dtype=torch.uint8, device=self.device)
        check(lib.tg_restart_seq_list_dev(C.byref(m), C.byref(r.graph.tcsr), C.byref(rs), cap, ptr(nids_cap), ptr(n_dev),
                                          ptr(t_dev), ptr(ws), ws.numel(), stream_ptr(self.device)), 'tg_restart_seq_list_dev')
        if self._pending is not None and getattr(self, '_gtab', None) is not None:
            ws2 = self._ws('gtab_rc', cap * (4 * self.memory_dim + 4) + 64)
            check(lib.tg_attn_gtab_rows(C.byref(m), cap, ptr(nids_cap), ptr(n_dev), ptr(ws2), ws2.numel(),
                                        stream_ptr(self.device)), 'tg_attn_gtab_rows(restart, captured)')

    @property
    def graph(self):
        return self.temporal_embedding_fn.graph

    @graph.setter
    def graph(self, new_obj):
        self.temporal_embedding_fn.graph = new_obj

    @property
    def device(self):
        return self.msg_memory.device

    def _apply(self, fn, *a, **kw):  # .to() / .cuda() move every tensor: pointers change
        eager, fused = self._pending is not None, self._fused is not None
        self._plists = None
        self._gtab = None
        self._ctab = None
        self._gtab_stamp = None
        self._struct_cache = None
        self._fused = None
        self._fused_l1 = None
        self._pending = None
        self._pending_stamp = None
        self._step_ws = {}
        out = super()._apply(fn, *a, **kw)
        if getattr(self, '_row_of', None) is not None:  # the row map of a partitioned model follows its tables
            self._row_of = fn(self._row_of)
        # derived tables follow the tensors to their new home: a model that streamed with eager updates / pre-multiplied
        # weights keeps doing so (silently falling back to the lazy forms would change nothing but the speed - and would
        # break engines that rely on the table, e.g. the partitioned multi-GPU layout)
        if eager:
            self.eager_updates()
        if fused and self.device.type == 'cuda':
            self.fuse_attention()
        return out

    def dropout_rng(self) -> Tensor:
        """device int64[2] = {seed, step counter} of the dropout mask generator (training only)"""
        t = self._step_ws.get('rng')
        if t is None or t.device != self.device:
            t = torch.tensor([torch.initial_seed() & (2 ** 63 - 1), 0], dtype=torch.int64, device=self.device)
            self._step_ws['rng'] = t
        return t

    def invalidate_struct(self):
        self._struct_cache = None

    def model_struct(self, layer: int = 0) -> TgModel:
        """tg_model view of this module's tensors (cached until tensors are re-homed).  layer: which attention layer's
        weights the struct carries (temporal_embedding_fn.fns[layer]; only the two-layer operator path asks for 1)."""
        if layer:
            base = self.model_struct()
            hit = getattr(self, '_struct_cache_l1', None)
            if hit is not None and hit[0] is base:
                return hit[1]
            m = TgModel.from_buffer_copy(base)  # ctypes structs with pointers do not copy.copy
            att = self.temporal_embedding_fn.fns[layer]
            lin = lambda l: TgLinear(ptr(l.weight), ptr(l.bias))
            mha = att.mha_fn
            m.attn_wq, m.attn_wk, m.attn_wv, m.attn_b_in = (ptr(mha.q_proj_weight), ptr(mha.k_proj_weight),
                                                           ptr(mha.v_proj_weight), ptr(mha.in_proj_bias))
            m.attn_out, m.attn_fc1, m.attn_fc2 = lin(mha.out_proj), lin(att.merger.fc1), lin(att.merger.fc2)
            f1 = getattr(self, '_fused_l1', None)
            m.attn_fused = ptr(f1) if f1 is not None else None
            self._struct_cache_l1 = (base, m)
            return m
        if self._struct_cache is not None:
            return self._struct_cache
        lin = lambda l: TgLinear(ptr(l.weight), ptr(l.bias))
        nul = TgLinear(None, None)
        fg, att = self.raw_feat_getter, self.temporal_embedding_fn.fns[0]
        for t in (fg.nfeats, fg.efeats):
            if t is not None and t.device != self.device and not (t.device.type == 'cpu' and t.is_pinned()):
                raise RuntimeError('feature tables must live on the model device or in pinned host memory '
                                   '(NumericalFeature(register_buffer=False) pins them)')
        tsfm1 = tsfm2 = nul
        if self.msg_tsfm_type == 'linear':
            tsfm1 = lin(self.msg_transform_fn.fn[1])
        elif self.msg_tsfm_type == 'mlp':
            tsfm1, tsfm2 = lin(self.msg_transform_fn.fn[1]), lin(self.msg_transform_fn.fn[4])
        gru = [None] * 4
        fc1 = fc2 = nul
        if self.mem_update_type == 'gru':
            c = self.right_mem_updater.cell
            gru = [ptr(c.weight_ih), ptr(c.weight_hh), ptr(c.bias_ih), ptr(c.bias_hh)]
        else:
            fc1, fc2 = lin(self.right_mem_updater.fn.fc1), lin(self.right_mem_updater.fn.fc2)
        L, R, S = self.left_memory, self.right_memory, self.msg_store
        mha = att.mha_fn
        m = TgModel(self.n_nodes, self.memory_dim, self.efeat_dim, self.n_neighbors, self.n_head,
                    0 if self.msg_src == 'left' else 1, 0 if self.upd_src == 'left' else 1,
                    _TSFM[self.msg_tsfm_type], _UPD[self.mem_update_type],
                    ptr(L.vals), ptr(L.update_ts), ptr(L.active_mask), ptr(R.vals), ptr(R.update_ts), ptr(R.active_mask),
                    ptr(S.node_msg_vals), ptr(S.node_msg_ts), ptr(S.has_msg_bits),
                    ptr(fg.nfeats), ptr(fg.efeats), ptr(self.time_encoder.basis_freq), ptr(self.time_encoder.phase),
                    tsfm1, tsfm2, gru[0], gru[1], gru[2], gru[3], fc1, fc2,
                    ptr(mha.q_proj_weight), ptr(mha.k_proj_weight), ptr(mha.v_proj_weight), ptr(mha.in_proj_bias),
                    lin(mha.out_proj), lin(att.merger.fc1), lin(att.merger.fc2),
                    ptr(self._fused) if self._fused is not None else None,
                    ptr(self._pending) if self._pending is not None else None,
                    ptr(self._row_of) if getattr(self, '_row_of', None) is not None else None,
                    ptr(self._gtab) if getattr(self, '_gtab', None) is not None else None,
                    ptr(self._ctab) if getattr(self, '_ctab', None) is not None else None)
        self._struct_cache = m
        return m

    # ---- eager updates (streaming with fixed parameters) ----------------------------------
    def eager_updates(self, enable: bool = True):
        """Streaming inference with FIXED parameters: h(t'+) of a node with a pending message is the same row
        every time the reference recomputes it (neither its mailbox row nor its memories change before the
        message is consumed), so `stream_step` / `launch_step` compute it ONCE, at the end of the batch that
        stores the message, into a [n_nodes, d] table; STEP 1-2 of later batches are then a gather of that
        table and the updater runs on the unique positive nodes of a batch instead of on every involved node
        with a pending message (C2: ~1 050 rows instead of ~4 200).  Same results as the lazy form.
        Every other method that touches state (restart, flush_msg, reset, contrast_learning, snapshots ...)
        is noticed and the table is rebuilt before the next eager step; after a parameter update or after
        writing to state tensors directly, call `invalidate_pending()`."""
        self._pending = None
        self._pending_stamp = None
        self._struct_cache = None
        if enable:
            self._pending = torch.zeros(self.msg_store.n, self.memory_dim, dtype=torch.float32, device=self.device)
        return self

    def partition_state(self, row_of: Tensor, n_rows: int):
        """PHYSICALLY partitioned state (multi-GPU; tiger_hip.h: tg_model.row_of): this process keeps `n_rows` rows of every
        state table - both memories, mailbox, has-message bitmap, eager-update table - instead of one per node.
        row_of[v] >= 0: the row of node v (rows the owner keeps for good); the caller re-points entries at arena rows for
        the nodes it pulls per batch.  Rows of nodes with row_of >= 0 are carried over from the current tables.  Node ids
        everywhere else (graph, batches, feature tables) stay global.  Only the partitioned engine's entry points address
        state by row: the model's own step / restart / flush methods must not be used on a partitioned model."""
        dev = self.device
        row_of = row_of.to(dev, torch.int32).contiguous()
        assert row_of.numel() == self.n_nodes and int(row_of.max()) < n_rows
        keep = torch.nonzero(row_of >= 0).flatten()
        rows = row_of[keep].long()
        prev = getattr(self, '_row_of', None)
        if prev is not None:  # already partitioned: the current tables are addressed by the CURRENT map
            src = prev.to(dev)[keep].long()
            if bool((src < 0).any()):
                raise ValueError('partition_state: a node gets a row that has none in the current partition')
            keep = src
        oldL, oldR, oldS, oldP = self.left_memory, self.right_memory, self.msg_store, self._pending
        L, R = Memory(n_rows, self.memory_dim).to(dev), Memory(n_rows, self.memory_dim).to(dev)
        S = MessageStoreNoGradLastOnly(n_rows, dim=self.raw_msg_dim).to(dev)
        for new, old in ((L, oldL), (R, oldR)):
            new.vals[rows] = old.vals[keep]
            new.update_ts[rows] = old.update_ts[keep]
            new.active_mask[rows] = old.active_mask[keep]
        S.node_msg_vals[rows] = oldS.node_msg_vals[keep]
        S.node_msg_ts[rows] = oldS.node_msg_ts[keep]
        has = rows[oldS._bits_of(keep).bool()]
        if has.numel():
            hip_ops.bitmap_mark(has, S.has_msg_bits, n_rows)
        self.left_memory, self.right_memory, self.msg_store = L, R, S
        self.msg_memory = L if self.msg_src == 'left' else R
        self.upd_memory = L if self.upd_src == 'left' else R
        self._row_of = row_of
        if oldP is not None:
            self._pending = torch.zeros(n_rows, self.memory_dim, dtype=torch.float32, device=dev)
            self._pending[rows] = oldP[keep]
        self._gtab = None
        self._ctab = None
        self._struct_cache = None
        self._pending_stamp = None
        self._touch()
        return self

    def invalidate_pending(self):
        self._pending_stamp = None
        self._gtab_stamp = None

    def _refuse_partitioned(self, what: str):
        """The model's own methods address state by NODE ID; on physically partitioned tables (partition_state: fewer rows
        than nodes) that would read and write other nodes' rows.  Only the partitioned engine's entry points (dist.py:
        embed-only lean step, serve / adopt / planned write-back / apply_messages on row lists) may run on such a model."""
        if getattr(self, '_row_of', None) is not None:
            raise RuntimeError(f'{what}: the model\'s state is physically partitioned (partition_state); only '
                               'www2023tiger_amd.dist.HipPartitionEngine may drive it')

    # ---- eager query rows (tiger_hip.h: tg_model.g_table) --------------------------------------------------------------
    def _gtab_wanted(self) -> bool:
        import os
        return (self._pending is not None and self._fused is not None and self.n_layers == 1
                and getattr(self, '_row_of', None) is None and os.environ.get('TG_GTAB', '1') != '0')

    def _sync_gtab(self):
        """The per-node table of folded attention queries G_v (one row per node; the fused eager step refreshes the rows of
        a batch's positive nodes itself): allocated when the model streams with eager updates AND pre-multiplied weights,
        rebuilt - all rows - whenever state or parameters changed outside that step."""
        if not self._gtab_wanted():
            if getattr(self, '_gtab', None) is not None:
                self._gtab = None
                self._ctab = None
                self._struct_cache = None
            return
        stamp = (self._state_stamp(), tuple(self._attn_stamp()), id(self._fused))
        if getattr(self, '_gtab', None) is not None and stamp == getattr(self, '_gtab_stamp', None):
            return
        if self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('eager query rows: run one step eagerly before capturing it into a graph')
        dev, n = self.device, self.msg_store.n
        nk = self.n_head * (2 * self.memory_dim + (self.efeat_dim if self.raw_feat_getter.efeats is not None else 0))
        if getattr(self, '_gtab', None) is None or self._gtab.shape != (n, nk):
            self._gtab = torch.empty(n, nk, dtype=torch.float32, device=dev)
            # ... and the centre rows c_v = e(v) + nfeat(v) beside them (tg_model.c_table; TG_CTAB=0: without)
            self._ctab = (torch.empty(n, self.memory_dim, dtype=torch.float32, device=dev)
                          if os.environ.get('TG_CTAB', '1') != '0' else None)
            self._struct_cache = None
        m = self.model_struct()
        chunk = 262144
        ws = self._ws('gtab', min(n, chunk) * (4 * self.memory_dim + 4) + 64)
        for lo in range(0, n, chunk):
            ids = torch.arange(lo, min(n, lo + chunk), dtype=torch.int64, device=dev)
            check(lib.tg_attn_gtab_rows(C.byref(m), ids.numel(), ptr(ids), None, ptr(ws), ws.numel(), stream_ptr(dev)),
                  'tg_attn_gtab_rows')
        self._gtab_stamp = stamp

    def _tables_follow_restart(self, nids: Tensor):
        """restart(nids) has just run on a model whose per-node tables were current: the restarted nodes have no pending
        message any more and new memories (tiger.py:603-609), nothing else changed - their centre / query rows are recomputed
        (tg_attn_gtab_rows) and the tables are declared current again, instead of a rebuild over every node at the next
        eager step."""
        if self._pending is None:
            return
        if getattr(self, '_gtab', None) is not None:
            m = self.model_struct()
            n = int(nids.numel())
            ws = self._ws('gtab_r', n * (4 * self.memory_dim + 4) + 64)
            check(lib.tg_attn_gtab_rows(C.byref(m), n, ptr(nids), None, ptr(ws), ws.numel(), stream_ptr(self.device)),
                  'tg_attn_gtab_rows(restart)')
            self._gtab_stamp = (self._state_stamp(), tuple(self._attn_stamp()), id(self._fused))
        self._pending_stamp = self._state_stamp()

    def _touch(self):
        self._state_version += 1

    def _state_stamp(self):
        """What the eager-update table is a function of, as far as the host can see it: the explicit version counters of
        the model and its state modules, torch's own version counters of the state tensors (any in-place torch operation
        on a memory / mailbox tensor bumps them - the library's kernels, which write through raw pointers, do not) and
        those of the updater / message-transform parameters (an optimizer step is an in-place update).  Writes through
        .data / raw pointers by third parties remain invisible: invalidate_pending() is theirs to call."""
        L, R, S = self.left_memory, self.right_memory, self.msg_store
        tv = (L.vals._version, L.update_ts._version, R.vals._version, R.update_ts._version, S.node_msg_vals._version,
              S.node_msg_ts._version, S.has_msg_bits._version)
        pl = self._param_lists()[0]
        pv = sum([p._version for p in pl]) + (getattr(self, '_param_epoch', 0) << 32)
        return (self._state_version, id(L), L._version_, id(R), R._version_, id(S), S._version_, tv, pv)

    def _param_lists(self):
        """(updater + message-transform parameters, attention + time-encoder parameters) as plain lists - walking the
        module tree on every step costs more host time than the step's launches; reset when the tensors are re-homed"""
        pl = getattr(self, '_plists', None)
        if pl is None:
            upd = [p for mod in (self.right_mem_updater, self.msg_transform_fn) for p in mod.parameters()]
            att = [p for mod in (self.temporal_embedding_fn, self.time_encoder) for p in mod.parameters()]
            pl = self._plists = (upd, att)
        return pl

    def train(self, mode: bool = True):
        """Entering train() mode starts a new parameter epoch: everything derived from parameters (pre-multiplied attention
        weights, the eager-update and per-node tables) is rebuilt before its next use.  Their stamps are torch's version
        counters, which the library's optimizer (tg_adam_step - www2023tiger_amd.optim.Adam, FusedTrainer, also inside replayed
        graphs) does not bump: it writes through raw pointers.  Whoever trains passes through train() first."""
        if mode:
            self._param_epoch = getattr(self, '_param_epoch', 0) + 1
        return super().train(mode)

    def _attn_stamp(self):
        """versions of everything the pre-multiplied weights are made of: attention + time encoder, and - the blob's tail
        for the split updater, W_hh W2 (csrc/tg_dense.h: GruTail) - the updater's parameters; + the parameter epoch (train())"""
        pl = self._param_lists()
        return [p._version for p in pl[1]] + [p._version for p in pl[0]] + [getattr(self, '_param_epoch', 0)]

    def _sync_pending(self):
        """Rebuild the table of precomputed updater rows if state changed outside the eager step:
        pending[v] = updater(upd_memory[v], tsfm(mailbox[v])) for every node with a pending message."""
        stamp = self._state_stamp()
        if stamp == self._pending_stamp:
            return
        if self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('eager updates: run one step eagerly before capturing it into a graph')
        dev = self.device
        m = self.model_struct()
        n_rows = self.msg_store.n  # (= n_nodes unless the state is physically partitioned: the bitmap is over rows)
        comp = hip_ops.unique_compact(self.msg_store.has_msg_bits, n_rows, n_rows)
        n = int(comp['count'].item())
        if n:
            ids = comp['ids'][:n].contiguous()
            ids32 = ids.to(torch.int32)
            err = hip_ops.new_err(dev)
            nbytes = int(lib.tg_apply_messages_workspace_bytes(C.byref(m), n))
            ws = self._ws('apply', nbytes)
            check(lib.tg_apply_messages(C.byref(m), ptr(ids), ptr(ids32), ptr(comp['count']), n, ptr(self._pending),
                                        ptr(err), ptr(ws), ws.numel(), stream_ptr(dev)), 'tg_apply_messages(pending)')
            hip_ops.raise_if_err(err)
        self._pending_stamp = stamp

    def fuse_attention(self, enable: bool = True):
        """Inference with FIXED parameters: pre-multiply the attention weights (tg_attn_fuse) so that the
        embedding runs three products instead of six.  Call again after any parameter update; training
        ignores the fused weights."""
        self._fused = None
        self._fused_l1 = None
        self._struct_cache = None
        if not enable:
            return self
        blobs = []
        for layer in range(self.n_layers):  # every attention layer has its own weights (fns[layer])
            m = self.model_struct(layer)
            n = int(lib.tg_attn_fused_floats(C.byref(m)))
            if n == 0:
                raise RuntimeError('tg_attn_fuse: unsupported model dimensions')
            fused = torch.empty(n, dtype=torch.float32, device=self.device)
            nbytes = int(lib.tg_attn_fuse_workspace_bytes(C.byref(m)))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            check(lib.tg_attn_fuse(C.byref(m), ptr(fused), ptr(ws), nbytes, stream_ptr(self.device)), 'tg_attn_fuse')
            blobs.append(fused)
        self._fused = blobs[0]
        self._fused_l1 = blobs[1] if len(blobs) > 1 else None
        self._fused_stamp = self._attn_stamp()
        self._struct_cache = None
        return self

    def _ws(self, key, nbytes) -> Tensor:
        t = self._step_ws.get(key)
        if t is None or t.numel() < nbytes or t.device != self.device:
            t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)
            self._step_ws[key] = t
        return t

    # ---- STEP 1-2 -----------------------------------------------------------------------
    def _consume(self, bitmap: Tensor, cap: int, err: Tensor):
        """involved / outdated decoding + reprs = right_memory[involved] with outdated rows
        replaced by updater(upd_memory, msg_fn(mailbox)) (tiger.py:208-221)."""
        m = self.model_struct()
        dev = self.device
        comp = hip_ops.unique_compact(bitmap, self.n_nodes, cap, and_bitmap=self.msg_store.has_msg_bits)  # bitmap is an input here
        reprs = torch.empty(cap, self.memory_dim, dtype=torch.float32, device=dev)
        s = stream_ptr(dev)
        check(lib.tg_mailbox_consume_gather(C.byref(m), ptr(comp['ids']), ptr(comp['count']), cap, ptr(reprs), s),
              'tg_mailbox_consume_gather')
        nbytes = int(lib.tg_apply_messages_workspace_bytes(C.byref(m), cap))
        ws = self._ws('apply', nbytes)
        check(lib.tg_apply_messages(C.byref(m), ptr(comp['and_ids']), ptr(comp['and_pos']), ptr(comp['and_count']),
                                    cap, ptr(reprs), ptr(err), ptr(ws), ws.numel(), s), 'tg_apply_messages')
        return comp, reprs

    def compute_messages(self, node_ids: Union[Tensor, np.ndarray, None] = None):
        """tiger.py:292-337 standalone form: (outdated ids, transformed messages, message ts)."""
        self._refuse_partitioned('compute_messages')
        outdated = self.msg_store.get_outdated_node_ids(node_ids).to(self.device)
        if len(outdated) == 0:
            return outdated, None, None
        last = self.msg_memory.update_ts[outdated]
        raw, ts = self.msg_aggregate_fn(outdated, last, self.msg_store.node_messages)
        if self.msg_src == 'left' and not (ts == last).all().item():
            raise ValueError("Messages' ts should be equal to last update ts when using left memory as msg source.")
        return outdated, self.msg_transform_fn(raw.detach()), ts

    # ---- the batch ----------------------------------------------------------------------
    def _train_forward(self, src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph, mutual: bool):
        """Training mode with autograd enabled: the whole iteration (collate, STEP 1-7, backward,
        write-back) runs as one tg_train_step; the returned losses carry an autograd node that
        hands the finished gradients to the parameters when `.backward()` is called."""
        from .training import TrainBuffers, hand_over
        self._touch()
        dev = self.device
        B = len(src_ids)
        key = ('train', B, mutual)
        tb = self._step_ws.get(key)
        if tb is None:
            tb = TrainBuffers(self, B, mutual=mutual)
            self._step_ws[key] = tb
        ts64 = getattr(computation_graph, 'ts64', None)  # the collator keeps the float64 event times
        if ts64 is None:
            ts64 = ts.to(dev).double()
        to = lambda x: x.to(dev).long()
        tb.sb.load(to(src_ids), to(dst_ids), to(neg_dst_ids), ts64, to(eids))
        cg_graph = getattr(computation_graph, 'graph', None)
        if cg_graph is None and hasattr(computation_graph, 'ts64'):
            raise NotImplementedError("training on device samples with strategy='recent_edges'")
        # every parameter owned by www2023tiger_amd.optim.Adam: nothing is read back in this iteration (the
        # invariant word of this batch is checked at the next one / flush_msg(), the live flags stay on the device)
        deferred = tb.err_host is not None and all(getattr(p, '_tg_deferred', False) for _, p, _ in tb.params)
        self._poll_train_errors()
        tb.launch(graph=cg_graph)  # sample where the collator sampled
        if deferred:
            tb.err_host[:1].copy_(tb.sb.err, non_blocking=True)
            tb.err_host[1:].copy_(tb.sb.counts[1:2], non_blocking=True)
            tb.err_event = torch.cuda.Event()
            tb.err_event.record()
        else:
            word = int(tb.sb.err.item())
            self.note_rows(int(tb.sb.counts[1].item()))
            if word:
                tb.sb.err.zero_()
                from .._lib import raise_invariants
                raise_invariants(word & 0xFFFFFFFF)
        grads = [(tb.grads[name], 1 if group == 2 else 0, group) for name, _, group in tb.params]
        losses = hand_over(tb.losses, grads, tb.flags if deferred else tb.flags.clone(),
                           [p for _, p, _ in tb.params], tb if deferred else None)
        return (losses, tb.sb.h[:2 * B].clone(), tb.pos_scores.clone(), tb.neg_scores.clone(),
                tb.sb.h_prev_left.clone(), tb.sb.h_prev_right.clone())

    def _poll_train_errors(self):
        """Deferred invariant check of the train steps launched without a read-back."""
        for key, tb in self._step_ws.items():
            ev = getattr(tb, 'err_event', None)
            if ev is None:
                continue
            ev.synchronize()
            tb.err_event = None
            word = int(tb.err_host[0].item())
            self.note_rows(int(tb.err_host[1].item()))
            if word:
                tb.sb.err.zero_()
                from .._lib import raise_invariants
                raise_invariants(word & 0xFFFFFFFF)

    def contrast_learning(self, src_ids: Tensor, dst_ids: Tensor, neg_dst_ids: Tensor, ts: Tensor, eids: Tensor,
                          computation_graph) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
        """tiger.py:174-290 -> (contrast_loss, h_left, pos_scores, neg_scores, h_prev_left, h_prev_right)"""
        self._refuse_partitioned('contrast_learning')
        if self.training and torch.is_grad_enabled():
            losses, *rest = self._train_forward(src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph, False)
            return (losses[0], *rest)
        with torch.no_grad():
            if (getattr(computation_graph, 'ts64', None) is not None
                    and getattr(computation_graph, 'graph', None) is not None and self._fused_eval_ok(computation_graph.graph)):
                return self._contrast_learning_fused_eval(src_ids, dst_ids, neg_dst_ids, eids, computation_graph)
            return self._contrast_learning_eval(src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph)

    def _fused_eval_ok(self, graph=None) -> bool:
        """does the one-call evaluation step (tg_train_step without gradient buffers) apply?  One or two layers, the
        recent-edges / recent-nodes / uniform strategies"""
        if self.hit_type == 'vec' and (2 * (self.nfeat_dim + self.n_neighbors)) % 4:
            return False  # the score head's pair rows must be float4-aligned
        strategy = getattr(graph if graph is not None else self.graph, 'strategy', 'recent_edges')
        return self.n_layers in (1, 2) and strategy in ('recent_edges', 'recent_nodes', 'uniform')

    def _contrast_learning_fused_eval(self, src_ids, dst_ids, neg_dst_ids, eids, computation_graph):
        """no_grad / eval(): collate, STEP 1-7 and the write-back as ONE device call (tg_train_step without
        gradient buffers).  The computation graph only contributes the float64 event times: the step
        samples the same neighbourhoods itself."""
        from .training import TrainBuffers
        dev = self.device
        B = len(src_ids)
        key = ('eval', B)
        tb = self._step_ws.get(key)
        if tb is None:
            tb = TrainBuffers(self, B, eval_only=True)
            self._step_ws[key] = tb
        to = lambda x: x.to(dev).long()
        tb.sb.load(to(src_ids), to(dst_ids), to(neg_dst_ids), computation_graph.ts64, to(eids))
        # the collator's graph, not model.graph: the reference embeds with the neighbourhoods the collator
        # sampled (e.g. warm-up batches are collated on the training graph while model.graph is the full one)
        self._poll_train_errors()  # invariant word of the previous batch (read back asynchronously, no stall)
        tb.launch(graph=getattr(computation_graph, 'graph', None))
        if tb.err_host is not None:
            tb.err_host[:1].copy_(tb.sb.err, non_blocking=True)
            tb.err_host[1:].copy_(tb.sb.counts[1:2], non_blocking=True)
            tb.err_event = torch.cuda.Event()
            tb.err_event.record()
        else:
            word = int(tb.sb.err.item())
            if word:
                tb.sb.err.zero_()
                from .._lib import raise_invariants
                raise_invariants(word & 0xFFFFFFFF)
        return (tb.losses[0].clone(), tb.sb.h[:2 * B].clone(), tb.pos_scores.clone(), tb.neg_scores.clone(),
                tb.sb.h_prev_left.clone(), tb.sb.h_prev_right.clone())

    def _contrast_learning_eval(self, src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph):
        cg = computation_graph
        self._touch()
        dev = self.device
        m = self.model_struct()
        s = stream_ptr(dev)
        bs, d = len(src_ids), self.memory_dim
        src_ids, dst_ids, neg_dst_ids, eids = (x.to(dev).long().contiguous() for x in
                                               (src_ids, dst_ids, neg_dst_ids, eids))
        ts = ts.to(dev).float().contiguous()
        pos = torch.cat([src_ids, dst_ids])
        batch_ids = torch.cat([pos, neg_dst_ids])
        ts2, ts3 = ts.repeat(2), ts.repeat(3)
        err = hip_ops.new_err(dev)
        K = self.n_neighbors
        cap = 3 * bs * (1 + K + (K * K if self.n_layers == 2 else 0))
        comp, reprs = self._consume(cg.bitmap, cap, err)  # STEP 1-2
        h_all = self.temporal_embedding_fn.compute_embedding_with_computation_graph(  # STEP 3
            reprs, batch_ids, ts3, cg, m if self.n_layers == 1 else self.model_struct, comp['rank'])
        upos, index = hip_ops.select_latest_nids(pos, ts2, self.n_nodes)  # dedup (tiger.py:232,419; memory.py:98)
        n_upos = torch.tensor([len(upos)], dtype=torch.int32, device=dev)
        check(lib.tg_consume_update_right(C.byref(m), ptr(upos), ptr(n_upos), len(upos), ptr(reprs), ptr(cg.bitmap),
                                          ptr(comp['rank']), ptr(err), s), 'tg_consume_update_right')  # STEP 4
        check(lib.tg_store_events(C.byref(m), bs, ptr(src_ids), ptr(dst_ids), ptr(ts), ptr(eids), ptr(upos),
                                  ptr(index), ptr(n_upos), ptr(err), s), 'tg_store_events')  # STEP 5
        h_prev_left = hip_ops.gather_rows(self.left_memory.vals, pos)  # restarter targets (tiger.py:248-251)
        h_prev_right = hip_ops.gather_rows(self.right_memory.vals, pos)
        h_left = h_all[:2 * bs]
        hip_ops.memory_scatter(self.left_memory.vals, self.left_memory.update_ts, self.left_memory.active_mask,
                               upos, h_all, ts2, src_index=index, check_past=True, err=err)  # STEP 6
        hip_ops.raise_if_err(err)
        # STEP 7 (tiger.py:257-288): score head, plain torch (outside the accelerated path)
        x, y, neg_y = h_all.reshape(3, bs, d)
        pos_scores, neg_scores = self._scores(x, y, neg_y, cg.hit_data)
        labels = torch.cat([torch.ones_like(pos_scores), torch.zeros_like(neg_scores)])
        loss = self.contrast_loss_fn(torch.cat([pos_scores, neg_scores]), labels)
        return loss, h_left, pos_scores, neg_scores, h_prev_left, h_prev_right

    def _scores(self, x, y, neg_y, hit_data):
        if self.hit_type in ('vec', 'bin', 'count'):
            sh, dh, nsh, ndh = hit_data
        if self.hit_type == 'vec':
            xp, yp = torch.cat([x, sh], 1), torch.cat([y, dh], 1)
            xn, yn = torch.cat([x, nsh], 1), torch.cat([neg_y, ndh], 1)
        elif self.hit_type in ('bin', 'count'):
            red = (lambda t: t.max(1).values.long()) if self.hit_type == 'bin' else (lambda t: t.sum(1).long())
            e = self.hit_embedding
            xp, yp, xn, yn = x + e(red(sh)), y + e(red(dh)), x + e(red(nsh)), neg_y + e(red(ndh))
        else:
            xp = xn = x
            yp, yn = y, neg_y
        return self.score_fn(xp, yp).squeeze(1), self.score_fn(xn, yn).squeeze(1)

    # ---- fused path ---------------------------------------------------------------------
    class StepBuffers:
        """Static device buffers of one batch size: inputs, outputs and the workspace of
        tg_stream_step; reused every step so the call sequence can be graph-captured."""

        def __init__(self, model: 'TIGE', B: int, want_prev: bool, resident=None, embed_only: bool = False,
                     h_out=None, h_new_out=None, want_h_new: bool = True, lean: bool = False, prefetch: bool = False,
                     debug_lists: bool = False, collate_only: bool = False):
            """resident = (src, dst, neg, ts64, eids) device tensors of the WHOLE stream: the
            step then reads batch [offset, offset+B) and advances `offset` on device.
            lean: the caller does not read `involved` nor counts[0:2] (tiger_hip.h: tg_step_io.lean) - an eager
            step then skips forming those sets; same results otherwise.
            prefetch (resident streams): the caller does not read the neighbour lists either (l1_* are not outputs then) - a
            lean eager step of a model with eager query rows then runs the NEXT batch's sampler and centres as riders of its
            own last launch (tiger_hip.h: tg_step_io.prefetch_state); same results otherwise.
            collate_only: buffers of a collate-only pass (tg_step_io.collate_only; the caller sets the flag): no embedding rows,
            the lists and the involved set stay in the workspace - a third of the allocations (a restart-mode evaluation
            pass builds sixteen such contexts)."""
            dev, d, K = model.device, model.memory_dim, model.n_neighbors
            self.B = B
            self.embed_only = embed_only
            i64 = dict(dtype=torch.int64, device=dev)
            self.offset = None
            if resident is None:
                self.src, self.dst, self.neg, self.eids = (torch.zeros(B, **i64) for _ in range(4))
                self.ts = torch.zeros(B, dtype=torch.float64, device=dev)
            else:
                self.src, self.dst, self.neg, self.ts, self.eids = resident
                assert self.ts.dtype == torch.float64 and all(t.is_contiguous() for t in resident)
                self.offset = torch.zeros(1, **i64)
            # h_out / h_new_out: caller-owned output rows (e.g. slices of an all-gather send buffer)
            if collate_only:
                self.h = self.l1_nids = self.l1_eids = self.l1_ts = self.involved = None
                want_prev = want_h_new = False
            else:
                self.h = h_out if h_out is not None else torch.zeros(3 * B, d, dtype=torch.float32, device=dev)
                self.l1_nids = torch.zeros(3 * B, K, **i64)
                self.l1_eids = torch.zeros(3 * B, K, **i64)
                self.l1_ts = torch.zeros(3 * B, K, dtype=torch.float32, device=dev)
                self.involved = torch.zeros(3 * B * (K + 1), **i64)
            self.counts = torch.zeros(4, dtype=torch.int32, device=dev)
            self.err = torch.zeros(1, dtype=torch.int32, device=dev)
            self.h_prev_left = torch.zeros(2 * B, d, dtype=torch.float32, device=dev) if want_prev else None
            self.h_prev_right = torch.zeros(2 * B, d, dtype=torch.float32, device=dev) if want_prev else None
            self.h_new = h_new_out if h_new_out is not None else (
                torch.zeros(2 * B, d, dtype=torch.float32, device=dev) if (embed_only and want_h_new) else None)
            m = model.model_struct()
            nbytes = int(lib.tg_stream_step_workspace_bytes2(C.byref(m), B, model.n_layers))
            if nbytes == 0:
                raise RuntimeError('tg_stream_step: unsupported model dimensions')
            self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)  # zero-filled: the step keeps it clean
            self.io = TgStepIo(B, ptr(self.src), ptr(self.dst), ptr(self.neg), ptr(self.ts), ptr(self.eids),
                               ptr(self.h), ptr(self.l1_nids), ptr(self.l1_eids), ptr(self.l1_ts), ptr(self.involved),
                               ptr(self.counts), ptr(self.h_prev_left), ptr(self.h_prev_right), ptr(self.err),
                               ptr(self.offset), 1 if resident is not None else 0, 1 if embed_only else 0, None,
                               ptr(self.h_new), 0 if embed_only else 1, 0)
            self.io.lean = 1 if lean else 0
            # collate prefetch (tiger_hip.h: tg_step_io.prefetch_state): a lean step over a resident stream runs the next
            # batch's sampler + centres on its own last launch; TIGE.launch_step keeps the promise the flag stands for
            self._pf_state = C.c_int32(0)
            self._pf_stamp = None
            if prefetch:
                if resident is None or embed_only:
                    raise ValueError('prefetch needs a resident stream and a full step')
                self.io.stream_len = int(self.src.numel())
                self.io.prefetch_state = C.addressof(self._pf_state)
                self.io.l1_nids = self.io.l1_eids = self.io.l1_ts = None
            # debug_lists: the neighbour lists of the batch a step CONSUMES, copied out before its last launch replaces them
            # (tiger_hip.h: tg_step_io.dbg_l1_*) - the collate-prefetch form has no other way to show them
            self.dbg_l1_nids = self.dbg_l1_eids = self.dbg_l1_ts = None
            if debug_lists:
                self.dbg_l1_nids = torch.zeros(3 * B, K, **i64)
                self.dbg_l1_eids = torch.zeros(3 * B, K, **i64)
                self.dbg_l1_ts = torch.zeros(3 * B, K, dtype=torch.float32, device=dev)
                self.io.dbg_l1_nids, self.io.dbg_l1_eids, self.io.dbg_l1_ts = (ptr(self.dbg_l1_nids), ptr(self.dbg_l1_eids),
                                                                               ptr(self.dbg_l1_ts))

        def attach_profiler(self, prof):
            self.io.profiler = prof

        def enable_lazy_restart(self, model: 'TIGER', trigger, force_list: bool = False):
            """The lazy-restart loop of train_self_supervised.py:152-163 inside the step (static restarter only;
            tiger_hip.h: tg_lazy_restart).  trigger[b] != 0 means the reference's `np.random.rand() < restart_prob`
            fired before batch b (the loop never fires before batch 0); the draws are the caller's, made up front,
            so a run is reproducible and the step stays free of host round trips.  counts[3] of every step is the
            number of nodes re-initialised by it.  force_list: the list form below for the static restarter too (a caller
            that drives another step function than launch_step, e.g. the resident evaluation pass)."""
            from .restarters import StaticRestarter
            r = getattr(model, 'restarter_fn', None)
            if r is None:
                raise NotImplementedError('the lazy-restart loop needs a model with a restarter (TIGER)')
            dev = model.device
            self.lazy_trigger = torch.as_tensor(trigger).to(dev, torch.uint8).contiguous()
            self.lazy_batch = torch.zeros(1, dtype=torch.int64, device=dev)
            self.lazy_restarting = torch.zeros(1, dtype=torch.int32, device=dev)
            self.lazy_uptodate = torch.zeros(hip_ops.bitmap_words(model.n_nodes), dtype=torch.int64, device=dev)
            if isinstance(r, StaticRestarter) and not force_list:  # the whole loop body runs inside the step
                self._lazy = TgLazyRestart(ptr(r.left_emb.weight), ptr(r.right_emb.weight), ptr(self.lazy_trigger),
                                           self.lazy_trigger.numel(), ptr(self.lazy_batch), ptr(self.lazy_restarting),
                                           ptr(self.lazy_uptodate), None, None)
                self.io.lazy = C.addressof(self._lazy)
                return self
            # Any other restarter (the SeqRestarter of the reference's default recipe, init_utils.py:55-58): the LIST form.
            # The loop's bookkeeping - trigger, uptodate / has-message bitmaps, involved & ~uptodate - runs on the device in
            # a collate-only pass; the host reads back ONE count, runs restarter + tg_restart_apply on the device-resident
            # list and then launches the step (TIGE.launch_step).  No node list crosses the host link, no Python sets.
            self.lazy_restarted = 0  # nodes re-initialised before the last step (the step's own counts[3] stays 0)
            cb = self.lazy_collate_context(model)
            self.lazy_list, self.lazy_tmin, self._lazy = cb.lazy_list, cb.lazy_tmin, cb._lazy
            self._lazy_collate = cb
            self._lazy_host = torch.zeros(2, dtype=torch.float32).pin_memory() if dev.type == 'cuda' else None
            return self

        def lazy_collate_context(self, model: 'TIGER'):
            """One collate-only pass of the list form (see enable_lazy_restart): step buffers of its own, its own list /
            earliest-time / count outputs, the loop variables (trigger, batch counter, restarting flag, up-to-date bitmap) of
            THIS buffer.  By default it reads the batch at this buffer's device-side offset without advancing it; a caller
            that runs passes ahead of the steps (eval_utils: the resident restart-mode pass) points `io.offset_dev` elsewhere."""
            dev = model.device
            cap = min(3 * self.B * (model.n_neighbors + 1), model.n_nodes)
            cb = TIGE.StepBuffers(model, self.B, False, resident=(self.src, self.dst, self.neg, self.ts, self.eids),
                                  collate_only=True)
            cb.lazy_list = torch.zeros(max(cap, 1), dtype=torch.int64, device=dev)
            cb.lazy_tmin = torch.zeros(1, dtype=torch.float32, device=dev)
            cb._lazy = TgLazyRestart(None, None, ptr(self.lazy_trigger), self.lazy_trigger.numel(), ptr(self.lazy_batch),
                                     ptr(self.lazy_restarting), ptr(self.lazy_uptodate), ptr(cb.lazy_list), ptr(cb.lazy_tmin))
            if self.offset is not None:  # the same batch as the step that follows: its device-side offset, not advanced
                cb.offset = self.offset
                cb.io.offset_dev = ptr(self.offset)
            cb.io.advance = 0
            cb.io.collate_only = 1
            cb.io.lazy = C.addressof(cb._lazy)
            return cb

        def load(self, src, dst, neg, ts, eids):
            self.src.copy_(src, non_blocking=True)
            self.dst.copy_(dst, non_blocking=True)
            self.neg.copy_(neg, non_blocking=True)
            self.ts.copy_(ts, non_blocking=True)
            self.eids.copy_(eids, non_blocking=True)

    def step_buffers(self, B: int, want_prev: bool = False, lean: bool = False) -> 'TIGE.StepBuffers':
        key = ('step', B, want_prev, lean)
        buf = self._step_ws.get(key)
        if buf is None:
            buf = TIGE.StepBuffers(self, B, want_prev, lean=lean)
            self._step_ws[key] = buf
        return buf

    def check_graph(self, graph):
        """the tables of the model and the bitmaps of a step are indexed by node id: a graph over fewer ids (e.g.
        Graph.from_data(train_data) when the training split lacks the highest ids) must be built with
        max_node_id = the largest id of the full data"""
        if graph.num_node != self.n_nodes:
            raise ValueError(f'graph covers {graph.num_node} node ids, the model {self.n_nodes}: build every graph with '
                             'max_node_id = the largest node id of the full data (init_utils.init_data does)')

    def prepare_pass(self, cb: 'TIGE.StepBuffers', graph=None):
        """A collate-only pass (StepBuffers.lazy_collate_context) must flag exactly what the step will involve: it samples with
        the graph's own strategy and, on a two-layer model, both hops (data_loader.py:105-131: `np_computation_graph_nodes`
        holds the nodes of every layer).  Call before each pass (the structs carry pointers that follow the model)."""
        strategy = getattr(self.graph if graph is None else graph, 'strategy', 'recent_edges')
        cb.io.strategy = {'recent_edges': 0, 'recent_nodes': 1, 'uniform': 2}.get(strategy, 0)
        if self.n_layers == 2:
            cb._inner = self.model_struct(1)
            cb.io.inner = C.addressof(cb._inner)

    def rows_bound(self) -> int:
        """Bound on the nodes with a pending message per batch handed to the library (tg_step_io.rows_hint):
        1.5 x the largest count read back so far, 0 while nothing has been read back.  Performance only - but a
        launch sized for one round that needs a second one costs more than the smaller blocks win, and the count
        grows while the graph behind the stream fills up (C2: 3200 at batch 20, 4550 at batch 120), hence the
        generous margin; the bound follows the counts as they are read back."""
        if self._pending is not None:  # eager updates: the updater runs on the unique positive nodes of a batch
            # (their count barely moves from batch to batch - C2: 1 040 .. 1 070 - and the updater kernels this selects
            # size their blocks from the LIVE count: a batch above the bound costs that batch some speed, nothing else)
            seen = getattr(self, '_pos_seen', 0)
            return int(1.03 * seen) + 1 if seen else 0
        seen = getattr(self, '_rows_seen', 0)
        return int(1.5 * seen) + 64 if seen else 0

    def note_rows(self, n_outdated: int, n_unique_pos: int = 0):
        self._rows_seen = max(getattr(self, '_rows_seen', 0), int(n_outdated))
        self._pos_seen = max(getattr(self, '_pos_seen', 0), int(n_unique_pos))

    def launch_step(self, buf: 'TIGE.StepBuffers'):
        """Enqueue collate + STEP 1-6 for the batch already in `buf` (no host sync)."""
        if not (buf.embed_only or buf.io.collate_only):
            self._refuse_partitioned('stream_step / launch_step (a full step)')
        strategy = getattr(self.graph, 'strategy', 'recent_edges')
        if strategy not in ('recent_edges', 'recent_nodes', 'uniform'):
            raise NotImplementedError(f"the fused step samples 'recent_edges', 'recent_nodes' or 'uniform'; strategy={strategy!r} "
                                      '(graph.py:104-110, alpha != 0) is not built')
        buf.io.strategy = {'recent_edges': 0, 'recent_nodes': 1, 'uniform': 2}[strategy]
        if strategy == 'uniform':  # the graph's RandomState lives on the device and is consumed in query order
            buf.io.mt_state = ptr(self.graph._mt_state())
        if self.n_layers == 2:  # the second attention layer's weights travel in a tg_model of their own
            buf._inner = self.model_struct(1)
            buf.io.inner = C.addressof(buf._inner)
        buf.io.rows_hint = self.rows_bound()
        if self._fused is not None and self._fused_stamp != self._attn_stamp():
            # an attention parameter was updated in place since the weights were pre-multiplied (an optimizer step, a
            # copy_): recompute them instead of embedding with stale products
            if self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('attention parameters changed since fuse_attention(): call it again before capturing')
            self.fuse_attention()
        self.check_graph(self.graph)
        g = self.graph.tcsr
        cb = getattr(buf, '_lazy_collate', None)
        if cb is not None:  # lazy-restart loop, list form (see StepBuffers.enable_lazy_restart)
            if self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('the lazy-restart loop with a sequence restarter reads one count back per batch: '
                                   'it cannot be captured into a graph (the static restarter runs inside the step)')
            m = self.model_struct()
            self.prepare_pass(cb, self.graph)
            check(lib.tg_stream_step(C.byref(m), C.byref(g), C.byref(cb.io), ptr(cb.ws), cb.ws.numel(),
                                     stream_ptr(self.device)), 'tg_stream_step(lazy restart list)')
            n = int(cb.counts[3].item())  # the one read-back of the loop
            buf.lazy_restarted = n
            if n:
                self.restart(buf.lazy_list[:n], buf.lazy_tmin.expand(n))
            buf.lazy_batch += 1
        if self._pending is not None and not buf.embed_only:
            self._sync_pending()
            # the per-node tables: every step without the in-step restart loop reads them; one with the loop only in the form
            # the LIBRARY reports for this very call (tg_stream_step_form: lean, static restarter, centre-row table, no
            # h_prev outputs, no compact copy ... - its decision, not a restatement of it here)
            if not buf.io.lazy:
                self._sync_gtab()
            elif self._gtab_wanted():
                if getattr(self, '_gtab', None) is None:
                    self._sync_gtab()  # (allocates: the form is a function of the struct the step will see)
                if lib.tg_stream_step_form(C.byref(self.model_struct()), C.byref(buf.io)) & 8:
                    self._sync_gtab()
        m = self.model_struct()
        pf = bool(buf.io.prefetch_state)
        if pf:  # is the collate part this buffer's previous step prefetched still the one this step needs?
            if buf._pf_state.value == 1 and buf._pf_stamp != self._prefetch_stamp(buf, g):
                if os.environ.get('TG_PF_DEBUG'):
                    print('prefetch discarded:', buf._pf_stamp, '->', self._prefetch_stamp(buf, g), flush=True)
                buf._pf_state.value = 2  # made, but for another state / offset / graph: the step discards it
            before = buf._pf_state.value
        self._step_serial = getattr(self, '_step_serial', 0) + 1
        check(lib.tg_stream_step(C.byref(m), C.byref(g), C.byref(buf.io), ptr(buf.ws), buf.ws.numel(),
                                 stream_ptr(self.device)), 'tg_stream_step')
        if pf:
            if self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                buf._pf_state.value = before  # nothing ran: the device is where it was before the capturing call
            buf._pf_stamp = self._prefetch_stamp(buf, g)
        if (buf.io.lazy and getattr(self, '_gtab', None) is not None
                and not (lib.tg_stream_step_form(C.byref(m), C.byref(buf.io)) & 8)):
            self._gtab_stamp = None  # the in-step restart loop re-initialised rows the tables did not follow

    def _prefetch_stamp(self, buf, g):
        """Everything the prefetched collate part of a batch is a function of, as far as the host can see it: the state
        (as for the eager-update table), that no other step ran on this model since, the stream offset tensor and the
        graph.  (Graph replays change none of these: a graph captured with the flag set on entry and exit keeps it
        valid by construction.)"""
        return (self._state_stamp(), getattr(self, '_step_serial', 0), buf.offset._version, id(buf.offset), C.addressof(g))

    @torch.no_grad()
    def stream_step(self, src, dst, neg, ts, eids, want_prev: bool = False, check_invariants: bool = True,
                    lean: bool = False):
        """Fused collate + STEP 1-6 (data_loader.py:77-131 + tiger.py:196-255).
        ts are the float64 event times.  Returns the StepBuffers (h = embeddings of
        cat[src,dst,neg]; rows [0,2B) are h_left).  lean: see StepBuffers."""
        dev = self.device
        to = lambda x, dt: torch.as_tensor(x).to(dev, dt)
        buf = self.step_buffers(len(src), want_prev, lean)
        buf.load(to(src, torch.int64), to(dst, torch.int64), to(neg, torch.int64), to(ts, torch.float64),
                 to(eids, torch.int64))
        self.launch_step(buf)
        if check_invariants:
            word = int(buf.err.item())
            cnt = buf.counts.tolist()
            self.note_rows(cnt[1], cnt[2])
            if word:
                buf.err.zero_()
                from .._lib import raise_invariants
                raise_invariants(word & 0xFFFFFFFF)
        return buf

    # ---- remaining reference methods -----------------------------------------------------
    @torch.no_grad()
    def update_right_memory(self, node_ids: Tensor, new_vals: Tensor, ts: Tensor):
        self.right_memory.set(node_ids, new_vals, ts)

    @torch.no_grad()
    def update_left_memory(self, node_ids: Tensor, new_vals: Tensor, ts: Tensor):
        node_ids, index = select_latest_nids(node_ids, ts, self.n_nodes)
        self.left_memory.set(node_ids, new_vals[index], ts[index])

    @torch.no_grad()
    def flush_msg(self):
        """tiger.py:444-455: consume every pending message into the right memory."""
        self._refuse_partitioned('flush_msg')
        self._poll_train_errors()
        self._touch()
        dev = self.device
        m = self.model_struct()
        err = hip_ops.new_err(dev)
        bits = self.msg_store.has_msg_bits
        comp, reprs = self._consume(bits, self.n_nodes, err)  # involved == outdated == all pending nodes
        n = int(comp['count'].item())
        if n:
            ids = comp['ids'][:n]
            mts = self.msg_store.node_msg_ts[ids]
            self.right_memory.set(ids, reprs[:n], mts)
            self.msg_store.clear(ids)
        hip_ops.raise_if_err(err)

    def load_state_dict(self, *args, **kwargs):
        """parameters (and memories) change under the derived tables: the pre-multiplied attention weights are
        recomputed and the eager-update table is rebuilt before its next use"""
        out = super().load_state_dict(*args, **kwargs)
        self._touch()
        if self._fused is not None:
            self.fuse_attention()
        return out

    def reset(self):
        self._touch()
        self.left_memory.clear()
        self.right_memory.clear()
        self.msg_store.clear()

    def save_memory_state(self):
        return (self.left_memory.clone(), self.right_memory.clone(), self.msg_store.clone())

    def load_memory_state(self, data):
        self._touch()
        self.left_memory, self.right_memory, self.msg_store = data
        self.msg_memory = self.left_memory if self.msg_src == 'left' else self.right_memory
        self.upd_memory = self.left_memory if self.upd_src == 'left' else self.right_memory
        self._struct_cache = None


class TIGER(TIGE):
    def __init__(self, *, raw_feat_getter, graph, restarter, n_neighbors: int = 20, n_layers: int = 2,
                 n_head: int = 2, dropout: float = 0.1, msg_src: str, upd_src: str, msg_tsfm_type: str = 'id',
                 mem_update_type: str = 'gru', tgn_mode: bool = True, msg_last_only: bool = True,
                 hit_type: str = 'vec'):
        super().__init__(raw_feat_getter=raw_feat_getter, graph=graph, n_neighbors=n_neighbors, n_layers=n_layers,
                         n_head=n_head, dropout=dropout, msg_src=msg_src, upd_src=upd_src,
                         msg_tsfm_type=msg_tsfm_type, mem_update_type=mem_update_type, tgn_mode=tgn_mode,
                         msg_last_only=msg_last_only, hit_type=hit_type)
        self.restarter_fn = restarter
        self.restarter_fn.model_struct_fn = self.model_struct
        self.restarter_fn.rng_fn = self.dropout_rng
        self.mutual_loss_fn = nn.MSELoss()

    def forward(self, *args, **kwargs):
        return self.contrast_and_mutual_learning(*args, **kwargs)

    def contrast_and_mutual_learning(self, src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph,
                                     contrast_only: bool = False):
        """tiger.py:547-592"""
        if self.training and torch.is_grad_enabled():
            losses, *_ = self._train_forward(src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph,
                                             mutual=not contrast_only)
            if contrast_only:
                return losses[0], torch.zeros((), dtype=torch.int64, device=losses.device)  # no host copy
            return losses[0], losses[1]
        with torch.no_grad():
            return self._contrast_and_mutual_eval(src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph,
                                                  contrast_only)

    def _contrast_and_mutual_eval(self, src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph,
                                  contrast_only: bool = False):
        contrast_loss, *_, h_prev_left, h_prev_right = self.contrast_learning(
            src_ids, dst_ids, neg_dst_ids, ts, eids, computation_graph)
        if contrast_only:
            return contrast_loss, torch.zeros((), dtype=torch.int64, device=contrast_loss.device)
        dev = self.device
        index = computation_graph.restart_data.index
        unique_nids = torch.cat([src_ids, dst_ids]).to(dev)[index]
        unique_ts = ts.to(dev).float().repeat(2)[index]
        sur_left, sur_right, _ = self.restarter_fn(unique_nids, unique_ts, computation_graph)
        targets = torch.cat([h_prev_left[index], h_prev_right[index]], 0)
        preds = torch.cat([sur_left, sur_right], 0)
        valid_rows = torch.where(~(targets == 0).all(1))[0]
        if len(valid_rows):
            mutual_loss = self.mutual_loss_fn(preds[valid_rows], targets[valid_rows])
        else:
            mutual_loss = torch.zeros((), dtype=torch.int64, device=dev)
        return contrast_loss, mutual_loss

    @torch.no_grad()
    def restart(self, nids: Tensor, ts: Tensor, mix: float = 0.):
        """tiger.py:594-609: fill both memories with the surrogate state."""
        self._refuse_partitioned('restart')
        if len(nids) == 0:
            return
        self._touch()
        dev = self.device
        nids = nids.to(dev).long().contiguous()
        h_left, h_right, prev_ts = self.restarter_fn(nids, ts.to(dev))
        if mix > 0:
            # the library's gather, not torch indexing: tables beyond 2^31 elements (10 M nodes x 256)
            h_left = mix * h_left + (1 - mix) * hip_ops.gather_rows(self.left_memory.vals, nids)
            h_right = mix * h_right + (1 - mix) * hip_ops.gather_rows(self.right_memory.vals, nids)
        m = self.model_struct()
        check(lib.tg_restart_apply(C.byref(m), nids.numel(), ptr(nids), ptr(h_left.contiguous()),
                                   ptr(h_right.contiguous()), ptr(prev_ts.contiguous()), stream_ptr(dev)),
              'tg_restart_apply')

    @property
    def graph(self):
        return self.temporal_embedding_fn.graph

    @graph.setter
    def graph(self, new_obj):
        self.temporal_embedding_fn.graph = new_obj
        self.restarter_fn.graph = new_obj
