"""Mirror of tiger/model/message_modules.py (the reachable classes)."""
from typing import Optional, Tuple

from torch import Tensor, nn

from .. import hip_ops
from .dense import hip_autograd, hip_inference, linear_autograd, linear_forward


class MessageFunction(nn.Module):
    def __init__(self, raw_msg_dim: int, out_msg_dim: Optional[int] = None):
        super().__init__()
        self.input_size = raw_msg_dim
        self.output_size = out_msg_dim

    def forward(self, raw_messages: Tensor) -> Tensor:
        raise NotImplementedError


class IdentityMessageFunction(MessageFunction):
    def __init__(self, raw_msg_dim: int, *args, **kwargs):
        super().__init__(raw_msg_dim, raw_msg_dim)

    def forward(self, raw_messages: Tensor) -> Tensor:
        return raw_messages


class LinearMessageFunction(MessageFunction):
    """Linear(4d -> 4d).  `fn` keeps the reference's Sequential layout (fn.1 is the Linear)
    so state_dict keys match; the engine applies it inside tg_apply_messages."""

    def __init__(self, raw_msg_dim: int, out_msg_dim: Optional[int] = None, dropout: float = 0.0):
        out_msg_dim = raw_msg_dim if out_msg_dim is None else out_msg_dim
        super().__init__(raw_msg_dim, out_msg_dim)
        self.fn = nn.Sequential(nn.Dropout(dropout), nn.Linear(raw_msg_dim, out_msg_dim))

    def forward(self, raw_messages: Tensor) -> Tensor:
        if hip_inference(raw_messages, self.fn[0], self.fn[1]):
            return linear_forward(self.fn[1], raw_messages)
        if hip_autograd(raw_messages, self.fn[1]):  # autograd / active dropout: the same kernels, backward included
            return linear_autograd(self.fn[1], self.fn[0](raw_messages))
        return self.fn(raw_messages)


class MLPMessageFunction(MessageFunction):
    """Linear(4d -> 2d), ReLU, Linear(2d -> 4d); fn.1 and fn.4 are the Linear layers."""

    def __init__(self, raw_msg_dim: int, out_msg_dim: Optional[int] = None, dropout: float = 0.0):
        out_msg_dim = raw_msg_dim if out_msg_dim is None else out_msg_dim
        super().__init__(raw_msg_dim, out_msg_dim)
        self.hidden_size = self.output_size // 2
        self.fn = nn.Sequential(nn.Dropout(dropout), nn.Linear(raw_msg_dim, self.hidden_size), nn.ReLU(),
                                nn.Dropout(dropout), nn.Linear(self.hidden_size, self.output_size))

    def forward(self, raw_messages: Tensor) -> Tensor:
        if hip_inference(raw_messages, self.fn[0], self.fn[1], self.fn[4]):
            return linear_forward(self.fn[4], linear_forward(self.fn[1], raw_messages, relu=True))
        if hip_autograd(raw_messages, self.fn[1], self.fn[4]):
            h = self.fn[2](linear_autograd(self.fn[1], self.fn[0](raw_messages)))
            return linear_autograd(self.fn[4], self.fn[3](h))
        return self.fn(raw_messages)


class LastMessageAggregatorNoGradLastOnly(nn.Module):
    """message_modules.py:150-160: gather the last raw message and its timestamp."""

    def __init__(self, raw_feat_getter, time_encoder):
        super().__init__()
        self.raw_feat_getter = raw_feat_getter
        self.time_encoder = time_encoder

    def forward(self, node_ids: Tensor, prev_ts: Tensor, node_msg: Tuple[Tensor, Tensor]) -> Tuple[Tensor, Tensor]:
        full_msgs, ts = hip_ops.gather_rows(node_msg[0], node_ids, node_msg[1])
        if (prev_ts > ts).any().item():
            raise ValueError('Messages happened later than memory updating.')
        return full_msgs, ts
