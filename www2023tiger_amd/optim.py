"""Adam for the reference's training loops (train_self_supervised.py:116,150,171:
`optim.Adam(model.parameters(), lr=lr)`, `zero_grad()`, `step()`), same constructor and methods.

Why not torch.optim.Adam: a parameter group that took no part in a batch (the updater on the very
first batch, the restarter without a mutual loss) has `.grad is None` in the reference, and torch's
Adam skips it, step count included.  Which groups were live is known on the device only, so handing
`None` gradients to torch needs a host synchronisation in every `backward()`.  This optimizer reads the
live flags ON the device (tg_adam_step): gradients are handed over as views of the step's flat
gradient buffer, nothing is read back, and the whole update is one launch instead of ~10 foreach
kernels over ~40 tensors.  The arithmetic is torch.optim.Adam's (bias-corrected, eps outside the
square root; amsgrad / weight decay are not used by the reference and not offered).

Parameters whose gradient came from ordinary autograd (anything outside the fused train step) are
updated with the same formula through torch ops."""
from typing import Dict, Tuple

import torch

from ._lib import TgAdamSeg, check, lib, ptr
from .hip_ops import stream_ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError('invalid Adam hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        for group in self.param_groups:
            for p in group['params']:
                # read by the train step's autograd node: every parameter it hands gradients to carries
                # the mark -> it may skip the host read-back of the live flags
                p._tg_deferred = True
        self._plans: Dict[tuple, tuple] = {}
        self._steps: Dict[torch.device, torch.Tensor] = {}

    def _moments(self, p):
        st = self.state[p]
        if 'exp_avg' not in st:
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _fused(self, group, ps):
        dev = ps[0].device
        flags = ps[0].grad._tg_group[0]
        key = (flags.data_ptr(),) + tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        plan = self._plans.get(key)
        if plan is None:
            segs = (TgAdamSeg * len(ps))()
            for i, p in enumerate(ps):
                if p.grad._tg_group[0] is not flags or not p.is_contiguous():
                    raise RuntimeError('gradients of one step must share their live flags')
                st = self._moments(p)
                segs[i] = TgAdamSeg(ptr(p), ptr(p.grad), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), p.numel(),
                                    p.grad._tg_group[1], 1.0)
            plan = (torch.frombuffer(bytearray(bytes(segs)), dtype=torch.uint8).to(dev), len(ps))
            self._plans[key] = plan
        steps = self._steps.get(dev)
        if steps is None:  # Adam's step count per parameter group of the model (idle groups do not tick)
            steps = self._steps[dev] = torch.zeros(4, dtype=torch.int32, device=dev)
        b1, b2 = group['betas']
        check(lib.tg_adam_step(ptr(plan[0]), plan[1], 4, ptr(flags), ptr(steps), float(group['lr']), b1, b2,
                               group['eps'], 1.0, stream_ptr(dev)), 'tg_adam_step')

    def _plain(self, group, ps):
        b1, b2 = group['betas']
        for p in ps:
            st = self._moments(p)
            st['step'] = st.get('step', 0) + 1
            g = p.grad
            st['exp_avg'].mul_(b1).add_(g, alpha=1 - b1)
            st['exp_avg_sq'].mul_(b2).addcmul_(g, g, value=1 - b2)
            c1, c2 = 1 - b1 ** st['step'], 1 - b2 ** st['step']
            denom = (st['exp_avg_sq'].sqrt() / c2 ** 0.5).add_(group['eps'])
            p.addcdiv_(st['exp_avg'], denom, value=-group['lr'] / c1)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            fused, plain = {}, []
            for p in group['params']:
                if p.grad is None:
                    continue
                tag = getattr(p.grad, '_tg_group', None)
                if tag is None:
                    plain.append(p)
                else:
                    fused.setdefault(id(tag[0]), []).append(p)
            for ps in fused.values():
                self._fused(group, ps)
            if plain:
                self._plain(group, plain)
        return loss
