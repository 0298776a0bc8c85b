"""Memory updaters (reference: tiger/model/update_modules.py).  Parameter containers only: the
attribute names (`cell`, `fn`) are what the reference's state_dict uses; inside the engine
the update runs as tg_apply_messages (fused float32-MFMA GRU, csrc/tg_gemm.hip).  Calling a
module directly runs the same kernels on dense rows."""
from torch import Tensor, nn

from . import dense
from .basic_modules import MergeLayer


class GRUUpdater(nn.Module):
    """h(t'+) = GRUCell(message, memory)   (update_modules.py:30-37)"""

    def __init__(self, msg_dim: int, memory_dim: int):
        super().__init__()
        self.msg_dim, self.memory_dim = msg_dim, memory_dim
        self.cell = nn.GRUCell(msg_dim, memory_dim)

    def forward(self, mem: Tensor, msg: Tensor, delta_ts: Tensor = None) -> Tensor:
        return dense.gru_forward(self.cell, msg, mem)


class MergeUpdater(nn.Module):
    """h(t'+) = MergeLayer(message, memory)   (update_modules.py:40-47)"""

    def __init__(self, msg_dim: int, memory_dim: int):
        super().__init__()
        self.msg_dim, self.memory_dim = msg_dim, memory_dim
        self.fn = MergeLayer(msg_dim, memory_dim, memory_dim, memory_dim)

    def forward(self, mem: Tensor, msg: Tensor, delta_ts: Tensor = None) -> Tensor:
        return self.fn(msg, mem)
