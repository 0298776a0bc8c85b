"""Training tail on device: one call per iteration of the reference's training loop
(train_self_supervised.py:143-171) - collate, STEP 1-7, backward, write-back
(`tg_train_step`) and `torch.optim.Adam` (`tg_adam_step`).

Two ways in:

* `TIGE.contrast_learning` / `TIGER.contrast_and_mutual_learning` in training mode return loss
  tensors that carry an autograd node: `loss.backward()` hands the gradients computed by the
  HIP backward pass to the parameters' `.grad`, so the reference's loop and any torch optimiser
  run unchanged.
* `FusedTrainer.step` keeps gradients and Adam state in flat device buffers and enqueues the
  whole iteration without touching the host (graph-capturable; the bench's training leg).
"""
import ctypes as C
from typing import List, Tuple

import torch
from torch import Tensor

from .._lib import (TgAdamSeg, TgLinear, TgModel, TgScoreParams, TgSeqRestarter, TgStepIo, TgTrainIo, check, lib,
                    ptr)
from ..hip_ops import stream_ptr

_HIT = {'none': 0, 'vec': 1, 'bin': 2, 'count': 3}


def contrast_parameters(model) -> List[Tuple[str, Tensor, int]]:
    """(state_dict name, parameter, Adam group) of everything the contrastive loss trains.
    Groups follow tg_train_io.flags: 0 always has a gradient, 1 only when the GRU ran."""
    att = model.temporal_embedding_fn.fns[0]
    mha = att.mha_fn
    pre = 'temporal_embedding_fn.fns.0.'
    out = [
        ('time_encoder.basis_freq', model.time_encoder.basis_freq, 0),
        ('time_encoder.phase', model.time_encoder.phase, 0),
    ]
    upd = model.right_mem_updater
    if model.mem_update_type == 'gru':
        out += [('right_mem_updater.cell.' + n, getattr(upd.cell, n), 1)
                for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    else:
        out += [(f'right_mem_updater.fn.{l}.{n}', getattr(getattr(upd.fn, l), n), 1)
                for l in ('fc1', 'fc2') for n in ('weight', 'bias')]
    if model.msg_tsfm_type != 'id':  # nn.Sequential indices of the Linear layers (message_modules.py:33-39,52)
        for idx in ((1,) if model.msg_tsfm_type == 'linear' else (1, 4)):
            lin = model.msg_transform_fn.fn[idx]
            out += [(f'msg_transform_fn.fn.{idx}.weight', lin.weight, 1), (f'msg_transform_fn.fn.{idx}.bias', lin.bias, 1)]
    out += [
        (pre + 'mha_fn.q_proj_weight', mha.q_proj_weight, 0),
        (pre + 'mha_fn.k_proj_weight', mha.k_proj_weight, 0),
        (pre + 'mha_fn.v_proj_weight', mha.v_proj_weight, 0),
        (pre + 'mha_fn.in_proj_bias', mha.in_proj_bias, 0),
        (pre + 'mha_fn.out_proj.weight', mha.out_proj.weight, 0),
        (pre + 'mha_fn.out_proj.bias', mha.out_proj.bias, 0),
        (pre + 'merger.fc1.weight', att.merger.fc1.weight, 0),
        (pre + 'merger.fc1.bias', att.merger.fc1.bias, 0),
        (pre + 'merger.fc2.weight', att.merger.fc2.weight, 0),
        (pre + 'merger.fc2.bias', att.merger.fc2.bias, 0),
        ('score_fn.fc1.weight', model.score_fn.fc1.weight, 0),
        ('score_fn.fc1.bias', model.score_fn.fc1.bias, 0),
        ('score_fn.fc2.weight', model.score_fn.fc2.weight, 0),
        ('score_fn.fc2.bias', model.score_fn.fc2.bias, 0),
    ]
    if model.hit_type in ('bin', 'count'):
        out.append(('hit_embedding.weight', model.hit_embedding.weight, 0))
    if model.n_layers == 2:  # the second attention layer (fns[1]: the neighbours' embeddings, temporal_agg_modules.py:57-66)
        att1 = model.temporal_embedding_fn.fns[1]
        out += attention_parameters('temporal_embedding_fn.fns.1.', att1)
    return out


def attention_parameters(pre, att):
    mha = att.mha_fn
    return [
        (pre + 'mha_fn.q_proj_weight', mha.q_proj_weight, 0),
        (pre + 'mha_fn.k_proj_weight', mha.k_proj_weight, 0),
        (pre + 'mha_fn.v_proj_weight', mha.v_proj_weight, 0),
        (pre + 'mha_fn.in_proj_bias', mha.in_proj_bias, 0),
        (pre + 'mha_fn.out_proj.weight', mha.out_proj.weight, 0),
        (pre + 'mha_fn.out_proj.bias', mha.out_proj.bias, 0),
        (pre + 'merger.fc1.weight', att.merger.fc1.weight, 0),
        (pre + 'merger.fc1.bias', att.merger.fc1.bias, 0),
        (pre + 'merger.fc2.weight', att.merger.fc2.weight, 0),
        (pre + 'merger.fc2.bias', att.merger.fc2.bias, 0),
    ]


def restarter_parameters(model) -> List[Tuple[str, Tensor, int]]:
    """Parameters the mutual loss trains (Adam group 2: they have a gradient only when some
    target row is non-zero, tiger.py:584-590)."""
    from .restarters import SeqRestarter
    r = model.restarter_fn
    if isinstance(r, SeqRestarter):
        named = [('time_encoder.basis_freq', r.time_encoder.basis_freq), ('time_encoder.phase', r.time_encoder.phase),
                 ('anony_emb.weight', r.anony_emb.weight), ('mha_fn.in_proj_weight', r.mha_fn.in_proj_weight),
                 ('mha_fn.in_proj_bias', r.mha_fn.in_proj_bias), ('mha_fn.out_proj.weight', r.mha_fn.out_proj.weight),
                 ('mha_fn.out_proj.bias', r.mha_fn.out_proj.bias), ('out_fn.weight', r.out_fn.weight),
                 ('out_fn.bias', r.out_fn.bias), ('merger.fc1.weight', r.merger.fc1.weight),
                 ('merger.fc1.bias', r.merger.fc1.bias), ('merger.fc2.weight', r.merger.fc2.weight),
                 ('merger.fc2.bias', r.merger.fc2.bias)]
    else:
        named = [('left_emb.weight', r.left_emb.weight), ('right_emb.weight', r.right_emb.weight)]
    return [('restarter_fn.' + n, p, 2) for n, p in named]


def seq_struct(r, g=None) -> TgSeqRestarter:
    """tg_seq_restarter over the SeqRestarter's parameters, or over same-named gradient views."""
    if g is None:
        return r._struct()
    t = lambda n: ptr(g['restarter_fn.' + n])
    return TgSeqRestarter(r.hist_len, r.n_head, t('time_encoder.basis_freq'), t('time_encoder.phase'),
                          t('anony_emb.weight'), t('mha_fn.in_proj_weight'), t('mha_fn.in_proj_bias'),
                          TgLinear(t('mha_fn.out_proj.weight'), t('mha_fn.out_proj.bias')),
                          TgLinear(t('out_fn.weight'), t('out_fn.bias')),
                          TgLinear(t('merger.fc1.weight'), t('merger.fc1.bias')),
                          TgLinear(t('merger.fc2.weight'), t('merger.fc2.bias')))


def score_struct(model, tensors=None) -> TgScoreParams:
    """tg_score_params over the model's score head, or over same-named gradient views."""
    t = tensors or {n: p for n, p, _ in contrast_parameters(model)}
    emb = t.get('hit_embedding.weight')
    return TgScoreParams(_HIT[model.hit_type], 0 if emb is None else emb.shape[0], ptr(emb),
                         TgLinear(ptr(t['score_fn.fc1.weight']), ptr(t['score_fn.fc1.bias'])),
                         TgLinear(ptr(t['score_fn.fc2.weight']), ptr(t['score_fn.fc2.bias'])))


def grads_struct(model, g, layer: int = 0) -> TgModel:
    """A tg_model whose parameter pointers address gradient buffers (sizes copied, state NULL).  layer: whose attention
    block the struct carries (1: the second layer's, tg_train_io.inner_grads)."""
    m = model.model_struct()
    pre = f'temporal_embedding_fn.fns.{layer}.'
    nul = TgLinear(None, None)
    lin = lambda stem: TgLinear(ptr(g[stem + '.weight']), ptr(g[stem + '.bias'])) if stem + '.weight' in g else nul
    gp = lambda name: ptr(g.get(name))
    return TgModel(m.n_nodes, m.d, m.d_e, m.n_neighbors, m.n_head, m.msg_src, m.upd_src, m.tsfm, m.upd_fn,
                   None, None, None, None, None, None, None, None, None, None, None,
                   ptr(g['time_encoder.basis_freq']), ptr(g['time_encoder.phase']),
                   lin('msg_transform_fn.fn.1'), lin('msg_transform_fn.fn.4'),
                   gp('right_mem_updater.cell.weight_ih'), gp('right_mem_updater.cell.weight_hh'),
                   gp('right_mem_updater.cell.bias_ih'), gp('right_mem_updater.cell.bias_hh'),
                   lin('right_mem_updater.fn.fc1'), lin('right_mem_updater.fn.fc2'),
                   ptr(g[pre + 'mha_fn.q_proj_weight']), ptr(g[pre + 'mha_fn.k_proj_weight']),
                   ptr(g[pre + 'mha_fn.v_proj_weight']), ptr(g[pre + 'mha_fn.in_proj_bias']),
                   TgLinear(ptr(g[pre + 'mha_fn.out_proj.weight']), ptr(g[pre + 'mha_fn.out_proj.bias'])),
                   TgLinear(ptr(g[pre + 'merger.fc1.weight']), ptr(g[pre + 'merger.fc1.bias'])),
                   TgLinear(ptr(g[pre + 'merger.fc2.weight']), ptr(g[pre + 'merger.fc2.bias'])), None)


def check_trainable(model):
    if model.msg_tsfm_type != 'id' and any(isinstance(l, torch.nn.Dropout) and l.p > 0 for l in model.msg_transform_fn.fn):
        raise NotImplementedError('dropout inside the message transform is not built (the reference never sets it)')
    if model.n_layers not in (1, 2):
        raise NotImplementedError('training on device supports n_layers 1 and 2')
    if model.n_layers == 2 and model.temporal_embedding_fn.fns[1].merger.dropout.p > 0:
        raise NotImplementedError('dropout inside the embedding merger is not built (the reference never sets it)')
    if getattr(model.graph, 'strategy', 'recent_edges') not in ('recent_edges', 'recent_nodes', 'uniform'):
        raise NotImplementedError("training on device samples with strategy 'recent_edges', 'recent_nodes' or 'uniform'")
    if model.temporal_embedding_fn.fns[0].merger.dropout.p > 0:
        raise NotImplementedError('dropout inside the embedding merger is not built (the reference never sets it)')
    dropout_p(model)


def dropout_p(model) -> float:
    """The single dropout probability of the model (the reference passes one --dropout to the score
    head, the embedding attention, and the SeqRestarter's attention and merger)."""
    ps = {float(model.score_fn.dropout.p), float(model.temporal_embedding_fn.fns[0].mha_fn.dropout)}
    r = getattr(model, 'restarter_fn', None)
    if r is not None and hasattr(r, 'mha_fn'):
        ps |= {float(r.mha_fn.dropout), float(r.merger.dropout.p)}
    if len(ps) != 1:
        raise NotImplementedError(f'one dropout probability for all sites is supported, got {sorted(ps)}')
    return ps.pop()


class TrainBuffers:
    """Static buffers of one batch size for tg_train_step: the step's inputs/outputs
    (a TIGE.StepBuffers), flat gradient storage with one view per parameter, losses, scores."""

    def __init__(self, model, B: int, resident=None, mutual: bool = False, eval_only: bool = False,
                 want_prev: bool = True, lean: bool = False, prefetch: bool = False):
        """mutual=True adds the restarter's mutual-learning loss (tiger.py:574-590) and its
        gradients; False is the reference's contrast_only (restart_prob == 0).
        eval_only=True: no gradient storage; the step computes embeddings, scores, loss and the
        write-back (the forward of tiger/eval_utils.py:29-48).
        want_prev=False / lean=True (evaluation only): no h_prev_left / h_prev_right outputs, no involved set - what an
        evaluation loop that only reads the scores needs; on a model that streams with eager updates and pre-multiplied
        weights the forward then takes the table-backed lean form of the streaming step (TIGE.StepBuffers);
        prefetch=True (resident stream, lean): the next batch's sampler + centres ride on the step's last launch."""
        if (not want_prev or lean) and not eval_only:
            raise ValueError('a training step hands h_prev_left / h_prev_right to the restarter: want_prev / lean are evaluation-only')
        check_trainable(model)
        model._refuse_partitioned('tg_train_step')
        dev = model.device
        self.model, self.B, self.mutual, self.eval_only = model, B, mutual and not eval_only, eval_only
        self.sb = model.StepBuffers(model, B, want_prev=want_prev, resident=resident, lean=lean, prefetch=prefetch)
        self.params = [] if eval_only else (contrast_parameters(model) + (restarter_parameters(model) if mutual else []))
        n = sum(p.numel() for _, p, _ in self.params)
        self.gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads, o = {}, 0
        self.n_contrast = 0  # gflat = [gradients of the contrast loss | gradients of the mutual loss]
        for name, p, group in self.params:
            self.grads[name] = self.gflat[o:o + p.numel()].view_as(p)
            self.grads[name]._tg_group = (None, group)  # completed below: (live flags, parameter group)
            o += p.numel()
            if group != 2:
                assert self.n_contrast == o - p.numel(), 'contrast parameters first'
                self.n_contrast = o
        self.losses = torch.zeros(2, dtype=torch.float32, device=dev)
        self.pos_scores = torch.zeros(B, dtype=torch.float32, device=dev)
        self.neg_scores = torch.zeros(B, dtype=torch.float32, device=dev)
        self.flags = torch.zeros(4, dtype=torch.int32, device=dev)
        for g in self.grads.values():
            g._tg_group = (self.flags, g._tg_group[1])  # read by www2023tiger_amd.optim.Adam
        # invariant word read back without stalling the loop (deferred mode): async copy + event
        self.err_host = torch.zeros(2, dtype=torch.int32).pin_memory() if dev.type == 'cuda' else None  # word, outdated count
        self.err_event = None
        self.rng = model.dropout_rng()  # dropout mask generator state {seed, step counter}, shared with restart()
        self.refresh()

    def refresh(self):
        """(Re)build the C structs; call again when parameters or state tensors were re-homed."""
        model, B = self.model, self.B
        self._score = score_struct(model)
        self._gmodel = None if self.eval_only else grads_struct(model, self.grads)
        self._gmodel1 = grads_struct(model, self.grads, 1) if (model.n_layers == 2 and not self.eval_only) else None
        self._inner = model.model_struct(1) if model.n_layers == 2 else None
        self._gscore = None if self.eval_only else score_struct(model, self.grads)
        m = model.model_struct()
        from .restarters import SeqRestarter
        kind, self._seq, self._gseq = 0, None, None
        if self.mutual:
            r = model.restarter_fn
            if isinstance(r, SeqRestarter):
                kind, self._seq, self._gseq = 1, seq_struct(r), seq_struct(r, self.grads)
            else:
                kind = 2
        nbytes = int(lib.tg_train_step_workspace_bytes2(C.byref(m), C.byref(self._score), kind,
                                                        C.addressof(self._seq) if self._seq is not None else None, B,
                                                        model.n_layers))
        if nbytes == 0:
            raise RuntimeError('tg_train_step: unsupported model configuration')
        if getattr(self, 'ws', None) is None or self.ws.numel() < nbytes:
            self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=model.device)
        io = TgTrainIo()
        C.memmove(C.addressof(io.step), C.addressof(self.sb.io), C.sizeof(TgStepIo))
        io.score = C.addressof(self._score)
        if self._inner is not None:  # --n_layers 2: the second attention layer's weights (and gradient buffers)
            io.step.inner = C.addressof(self._inner)
            if self._gmodel1 is not None:
                io.inner_grads = C.addressof(self._gmodel1)
        if not self.eval_only:
            io.grads = C.addressof(self._gmodel)
            io.score_grads = C.addressof(self._gscore)
        io.losses, io.pos_scores, io.neg_scores = ptr(self.losses), ptr(self.pos_scores), ptr(self.neg_scores)
        io.flags = ptr(self.flags)
        io.dropout_p = 0.0 if self.eval_only else dropout_p(model)
        io.rng = ptr(self.rng)
        io.restarter = kind
        if kind == 1:
            io.seq, io.seq_grads = C.addressof(self._seq), C.addressof(self._gseq)
        elif kind == 2:
            r = model.restarter_fn
            io.static_left, io.static_right = ptr(r.left_emb.weight), ptr(r.right_emb.weight)
            io.static_left_grad = ptr(self.grads['restarter_fn.left_emb.weight'])
            io.static_right_grad = ptr(self.grads['restarter_fn.right_emb.weight'])
        self.io = io

    def launch(self, zero_grads: bool = True, graph=None):
        """Enqueue forward + STEP 7 + backward + write-back for the batch in `sb` (no host sync).
        `graph`: the graph the neighbourhoods AND the mutual loss's histories are sampled from - the
        collator's graph, as in the reference (tiger.py:579-581 reads the collated restart data);
        default model.graph."""
        model = self.model
        self._grads_taken = False
        if zero_grads and not self.eval_only:
            self.gflat.zero_()
        m = model.model_struct()
        self.io.step.rows_hint = model.rows_bound()
        graph = model.graph if graph is None else graph
        # the neighbourhoods follow the graph's strategy (graph.py:94-148); the hit windows are recent-edges lists always
        strategy = getattr(graph, 'strategy', 'recent_edges')
        self.io.step.strategy = {'recent_edges': 0, 'recent_nodes': 1, 'uniform': 2}[strategy]
        self.io.step.mt_state = ptr(graph._mt_state()) if strategy == 'uniform' else None
        model.check_graph(graph)
        if self.eval_only and model._pending is not None:
            # a model that streams with eager updates: the evaluation step IS the streaming step (+ STEP 7) and keeps the
            # per-node tables current itself - same protocol as TIGE.launch_step (tables synchronised before, no touch)
            if model._fused is not None and model._fused_stamp != model._attn_stamp():
                model.fuse_attention()
            model._sync_pending()
            model._sync_gtab()
            m = model.model_struct()
            g = graph.tcsr
            buf = self.sb
            pf = bool(self.io.step.prefetch_state)
            if pf:  # is the collate part this buffer's previous step prefetched still the one this step needs?
                if buf._pf_state.value == 1 and buf._pf_stamp != model._prefetch_stamp(buf, g):
                    buf._pf_state.value = 2  # made, but for another state / offset / graph: the step discards it
                before = buf._pf_state.value
            model._step_serial = getattr(model, '_step_serial', 0) + 1
            check(lib.tg_train_step(C.byref(m), C.byref(g), C.byref(self.io), ptr(self.ws), self.ws.numel(),
                                    stream_ptr(model.device)), 'tg_train_step')
            if pf:
                if model.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                    buf._pf_state.value = before  # nothing ran: the device is where it was before the capturing call
                buf._pf_stamp = model._prefetch_stamp(buf, g)
            if (self.io.step.lazy and getattr(model, '_gtab', None) is not None
                    and not (lib.tg_stream_step_form(C.byref(m), C.byref(self.io.step)) & 8)):
                model._gtab_stamp = None  # the in-step restart loop re-initialised rows the tables did not follow
            return
        model._touch()  # state changes outside the eager streaming step
        g = graph.tcsr
        check(lib.tg_train_step(C.byref(m), C.byref(g), C.byref(self.io), ptr(self.ws), self.ws.numel(),
                                stream_ptr(model.device)), 'tg_train_step')


class _HandOver(torch.autograd.Function):
    """Connects the losses computed by tg_train_step to autograd: backward returns the gradients
    the HIP backward pass already produced, scaled by the incoming loss gradient.

    Deferred mode (every parameter is owned by www2023tiger_amd.optim.Adam): no host read-back.  The flat
    gradient buffer is scaled in place and each parameter's `.grad` becomes its view of that buffer
    (stable addresses: the optimizer's launch plan is built once); idle groups keep zero gradients
    and are skipped by the optimizer on the device."""

    @staticmethod
    def forward(ctx, losses, grads, flags, buf, *params):
        ctx.grads, ctx.flags, ctx.buf, ctx.params = grads, flags, buf, params
        return losses.clone()

    @staticmethod
    def backward(ctx, g_losses):
        if ctx.buf is not None:
            tb = ctx.buf
            # the gradients of this step live in ONE buffer that backward() scales in place and hands over as the
            # .grad views: that REPLACES .grad (the reference's loop zero_grad()s every iteration, so nothing is
            # lost there); accumulating over several batches or a second backward() of the same step would silently
            # give wrong gradients, so both are refused - accumulate with torch.optim.Adam (non-deferred hand-over)
            if getattr(tb, '_grads_taken', False):
                raise RuntimeError('backward() was already run for this training step: its gradient buffer is '
                                   'handed over in place and cannot be back-propagated twice')
            for p, (g, _, _) in zip(ctx.params, ctx.grads):
                if p.grad is not None and p.grad.data_ptr() != g.data_ptr():
                    raise RuntimeError('gradient accumulation is not supported with www2023tiger_amd.optim.Adam '
                                       '(call zero_grad() every iteration, or use torch.optim.Adam)')
            tb._grads_taken = True
            tb.gflat[:tb.n_contrast].mul_(g_losses[0])
            if tb.n_contrast < tb.gflat.numel():
                tb.gflat[tb.n_contrast:].mul_(g_losses[1])
            for p, (g, _, _) in zip(ctx.params, ctx.grads):
                p.grad = g
            return (None,) * (4 + len(ctx.params))
        # grads: (gradient view, index of the loss it belongs to, parameter group).  A group whose flag
        # is 0 took no part in the graph this step: torch would leave .grad None (and Adam would skip
        # the parameter, step count included), so None is returned for it.
        live = ctx.flags.tolist()
        out = tuple((g * g_losses[k]) if live[grp] else None for g, k, grp in ctx.grads)
        return (None, None, None, None) + out


def hand_over(losses: Tensor, grads: List[Tuple[Tensor, int, int]], flags: Tensor, params: List[Tensor],
              deferred_buf=None) -> Tensor:
    return _HandOver.apply(losses, grads, flags, deferred_buf, *params)


class FusedTrainer:
    """The training loop's device work with no host round trip per iteration: tg_train_step then
    tg_adam_step over flat parameter-gradient / moment buffers."""

    def __init__(self, model, B: int, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, resident=None,
                 mutual: bool = False, mutual_coef: float = 1.0, process_group=None, world_size: int = 1):
        """process_group / world_size > 1: the reference's data-parallel recipe
        (train_self_supervised_ddp.py:145-146: every rank trains its own time chunk, gradients are
        averaged, lr is scaled by the caller).  All gradients live in ONE flat buffer, so the
        synchronisation is a single all-reduce over RCCL between the backward pass and Adam."""
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.mutual_coef = mutual_coef
        self.process_group, self.world_size = process_group, world_size
        self.buf = TrainBuffers(model, B, resident=resident, mutual=mutual)
        dev = model.device
        n = self.buf.gflat.numel()
        self.m1 = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m2 = torch.zeros(n, dtype=torch.float32, device=dev)
        segs = (TgAdamSeg * len(self.buf.params))()
        o = 0
        for i, (name, p, group) in enumerate(self.buf.params):
            k = p.numel()
            segs[i] = TgAdamSeg(ptr(p), ptr(self.buf.grads[name]), self.m1[o:o + k].data_ptr(),
                                self.m2[o:o + k].data_ptr(), k, group, mutual_coef if group == 2 else 1.0)
            o += k
        raw = bytes(segs)
        self.segs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.n_segs = len(self.buf.params)
        self.steps = torch.zeros(4, dtype=torch.int32, device=dev)

    def load(self, src, dst, neg, ts, eids):
        dev = self.model.device
        to = lambda x, dt: torch.as_tensor(x).to(dev, dt)
        self.buf.sb.load(to(src, torch.int64), to(dst, torch.int64), to(neg, torch.int64), to(ts, torch.float64),
                         to(eids, torch.int64))

    def enable_lazy_restart(self, trigger):
        """The lazy-restart loop of the training script (train_self_supervised.py:152-163) in front of every iteration:
        trigger[b] != 0 - the caller's pre-drawn `np.random.rand() < restart_prob` of batch b (never before batch 0) - forgets
        who is up to date and drops every pending message; from then on every batch re-initialises its involved nodes that are
        not up to date with TIGER.restart at the batch's earliest time (the restarter in train() mode, as the reference
        calls it).  StaticRestarter: the whole loop body runs inside the step (no host round trip, capturable).  Any other
        restarter: the bookkeeping runs on the device in a collate-only pass, the host reads ONE count per iteration and
        calls the restarter on the device-resident list (not capturable).  The up-to-date set starts empty, `restarting`
        False - one call per epoch, as the reference re-creates both per epoch."""
        sb = self.buf.sb
        sb.enable_lazy_restart(self.model, trigger)
        if getattr(sb, '_lazy_collate', None) is None:
            self.buf.refresh()  # (the step's io is a copy of the buffer's: it now carries the lazy-restart block)
        self.restarted = 0  # nodes re-initialised before the last iteration (list form)
        return self

    def launch(self, graph=None):
        sb = self.buf.sb
        cb = getattr(sb, '_lazy_collate', None)
        if cb is not None:  # list form: pass -> one count -> TIGER.restart on the device-resident list -> the step
            model = self.model
            if model.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('the lazy-restart loop with a sequence restarter reads one count back per iteration: '
                                   'it cannot be captured into a graph (the static restarter runs inside the step)')
            g = (model.graph if graph is None else graph).tcsr
            m = model.model_struct()
            model.prepare_pass(cb, model.graph if graph is None else graph)
            check(lib.tg_stream_step(C.byref(m), C.byref(g), C.byref(cb.io), ptr(cb.ws), cb.ws.numel(),
                                     stream_ptr(model.device)), 'tg_stream_step(lazy restart list)')
            n = self.restarted = int(cb.counts[3].item())
            if n:  # (one library call for the SeqRestarter without dropout, TIGER.restart otherwise)
                model.restart_list(sb.lazy_list[:n], sb.lazy_tmin)
            sb.lazy_batch += 1
        self.buf.launch(graph=graph)
        gscale = 1.0
        if self.world_size > 1:
            import torch.distributed as dist
            # a group is "live" if it produced a gradient on ANY rank (DDP semantics: an all-reduced
            # gradient is never None), so the flags are max-reduced together with the gradient sum
            dist.all_reduce(self.buf.gflat, group=self.process_group)
            dist.all_reduce(self.buf.flags, op=dist.ReduceOp.MAX, group=self.process_group)
            gscale = 1.0 / self.world_size
        check(lib.tg_adam_step(ptr(self.segs), self.n_segs, 4, ptr(self.buf.flags), ptr(self.steps), self.lr,
                               self.betas[0], self.betas[1], self.eps, gscale, stream_ptr(self.model.device)),
              'tg_adam_step')

    def step(self, src, dst, neg, ts, eids):
        self.load(src, dst, neg, ts, eids)
        self.launch()
        return self.buf.losses
