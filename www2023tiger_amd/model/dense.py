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


def gru_forward(cell: nn.GRUCell, x: Tensor, h: Tensor) -> Tensor:
    x, h = x.contiguous().float(), h.contiguous().float()
    out = torch.empty_like(h)
    check(lib.tg_gru_fwd(x.shape[0], ptr(x), x.shape[1], ptr(h), h.shape[1], ptr(cell.weight_ih), ptr(cell.weight_hh),
                         ptr(cell.bias_ih), ptr(cell.bias_hh), ptr(out), stream_ptr(x.device)), 'tg_gru_fwd')
    return out
