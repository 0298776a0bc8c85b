"""Mirror of tiger/model/time_encoding.py."""
import numpy as np
import torch
from torch import Tensor, nn

from .. import hip_ops


class TimeEncode(nn.Module):
    """TGAT harmonic time encoding cos(fl32(t * w) + phi); forward runs tg_time_encode
    (inference path: no autograd graph is recorded)."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.basis_freq = nn.Parameter(torch.from_numpy(1 / 10 ** np.linspace(0, 9, dim)).float())
        self.phase = nn.Parameter(torch.zeros(dim).float())

    def forward(self, ts: Tensor) -> Tensor:
        return hip_ops.time_encode(ts, self.basis_freq.detach(), self.phase.detach())
