"""Standalone dense entry points (float32 MFMA kernels of csrc/tg_gemm.hip)."""
import ctypes as C

import torch
from torch import Tensor, nn

from .._lib import TgLinear, check, lib, ptr
from ..hip_ops import stream_ptr


def hip_inference(x: Tensor, dropout: nn.Dropout, *layers: nn.Linear) -> bool:
    """May a module's forward run on the library's kernels?  Inference only (no autograd graph to build, no active
    dropout mask), a 2-D float32 input on the GPU, widths the kernels take (multiples of four floats)."""
    if torch.is_grad_enabled() or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32:
        return False
    if dropout.training and dropout.p > 0:
        return False
    return all(l.in_features % 4 == 0 and l.bias is not None and l.weight.is_cuda for l in layers)


def linear_forward(layer: nn.Linear, x: Tensor, relu: bool = False) -> Tensor:
    x = x.contiguous().float()
    n, in_f = x.shape
    out = torch.empty(n, layer.out_features, dtype=torch.float32, device=x.device)
    lin = TgLinear(ptr(layer.weight), ptr(layer.bias))
    check(lib.tg_linear_fwd(n, ptr(x), in_f, C.byref(lin), layer.out_features, 1 if relu else 0, ptr(out),
                            stream_ptr(x.device)), 'tg_linear_fwd')
    return out


class _LinearFn(torch.autograd.Function):
    """nn.Linear on the library's kernels WITH autograd: forward tg_linear_fwd, backward tg_linear_bwd (dx = dy W,
    dw = dy^T x, db = column sums) - the operator path under autograd stays on one backend."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w, b = x.contiguous(), w.contiguous(), b.contiguous()
        n, in_f = x.shape
        out_f = w.shape[0]
        out = torch.empty(n, out_f, dtype=torch.float32, device=x.device)
        lin = TgLinear(ptr(w), ptr(b))
        check(lib.tg_linear_fwd(n, ptr(x), in_f, C.byref(lin), out_f, 0, ptr(out), stream_ptr(x.device)), 'tg_linear_fwd')
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous().float()
        n, in_f = x.shape
        out_f = w.shape[0]
        need_x, need_w, need_b = ctx.needs_input_grad
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(w) if (need_w or need_b) else None  # (db comes out of dw's pass)
        db = torch.empty(out_f, dtype=torch.float32, device=x.device) if need_b else None
        if n == 0:
            return (None if dx is None else dx.zero_(), None if not need_w else dw.zero_(), None if db is None else db.zero_())
        nbytes = int(lib.tg_linear_bwd_workspace_bytes(in_f, out_f)) if dw is not None else 0
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
        check(lib.tg_linear_bwd(n, ptr(x), in_f, ptr(w), out_f, ptr(dy), ptr(dx), ptr(dw), ptr(db), ptr(ws), ws.numel(),
                                stream_ptr(x.device)), 'tg_linear_bwd')
        return dx, (dw if need_w else None), db


def hip_autograd(x: Tensor, *layers: nn.Linear) -> bool:
    """May a module's forward run on the library's kernels under autograd?  A 2-D float32 input on the GPU, input widths
    that are multiples of four floats (output widths are padded where they are not), biases present."""
    if not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32:
        return False
    return all(l.in_features % 4 == 0 and l.bias is not None and l.weight.is_cuda for l in layers)


def linear_autograd(layer: nn.Linear, x: Tensor) -> Tensor:
    """layer(x) through _LinearFn; an output width that is no multiple of four (the score head's d -> 1) runs on weights
    padded with zero rows - the padding and the slice are differentiable torch views / copies, the products the library's."""
    w, b = layer.weight, layer.bias
    out_f = w.shape[0]
    pad = (-out_f) % 4
    if pad:
        w = torch.cat([w, w.new_zeros(pad, w.shape[1])], 0)
        b = torch.cat([b, b.new_zeros(pad)], 0)
    y = _LinearFn.apply(x, w, b)
    return y[:, :out_f] if pad else y


def gru_forward(cell: nn.GRUCell, x: Tensor, h: Tensor) -> Tensor:
    x, h = x.contiguous().float(), h.contiguous().float()
    out = torch.empty_like(h)
    check(lib.tg_gru_fwd(x.shape[0], ptr(x), x.shape[1], ptr(h), h.shape[1], ptr(cell.weight_ih), ptr(cell.weight_hh),
                         ptr(cell.bias_ih), ptr(cell.bias_hh), ptr(out), stream_ptr(x.device)), 'tg_gru_fwd')
    return out
