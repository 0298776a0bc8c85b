"""Mirror of tiger/model/update_modules.py.  The modules are parameter containers with
the reference's names/initialisers; inside the engine they run as tg_apply_messages
(fused f32-MFMA GRU, csrc/tg_gemm.hip).  `forward` is the standalone dense form."""
import torch
from torch import Tensor, nn

from .basic_modules import MergeLayer


class UpdateModule(nn.Module):
    def __init__(self, msg_dim, memory_dim):
        super().__init__()
        self.msg_dim = msg_dim
        self.memory_dim = memory_dim

    def forward(self, mem: Tensor, msg: Tensor, delta_ts: Tensor) -> Tensor:
        raise NotImplementedError


class GRUUpdater(UpdateModule):
    def __init__(self, msg_dim, memory_dim):
        super().__init__(msg_dim, memory_dim)
        self.cell = nn.GRUCell(input_size=self.msg_dim, hidden_size=self.memory_dim)

    def forward(self, mem: Tensor, msg: Tensor, delta_ts: Tensor = None) -> Tensor:
        from .dense import gru_forward
        return gru_forward(self.cell, msg, mem)


class MergeUpdater(UpdateModule):
    def __init__(self, msg_dim, memory_dim):
        super().__init__(msg_dim, memory_dim)
        self.fn = MergeLayer(msg_dim, memory_dim, memory_dim, memory_dim)

    def forward(self, mem: Tensor, msg: Tensor, delta_ts: Tensor = None) -> Tensor:
        return self.fn(msg, mem)
