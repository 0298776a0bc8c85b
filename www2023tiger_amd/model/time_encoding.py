"""TGAT harmonic time encoding (reference: tiger/model/time_encoding.py)."""
import numpy as np
import torch
from torch import Tensor, nn

from .. import hip_ops


class TimeEncode(nn.Module):
    """out[..., j] = cos(fl32(t * basis_freq[j]) + phase[j]), frequencies 10^-linspace(0, 9, dim),
    zero phase at initialisation.  The forward is tg_time_encode (no autograd graph: gradients of
    the two parameters come from the training step's backward kernels)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        freq = np.power(10.0, -np.linspace(0, 9, dim))
        self.basis_freq = nn.Parameter(torch.as_tensor(freq, dtype=torch.float32))
        self.phase = nn.Parameter(torch.zeros(dim, dtype=torch.float32))

    def forward(self, ts: Tensor) -> Tensor:
        return hip_ops.time_encode(ts, self.basis_freq.detach(), self.phase.detach())
