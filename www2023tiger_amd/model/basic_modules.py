"""MergeLayer (reference: tiger/model/basic_modules.py:5-19); the node-classification MLP of that
file belongs to a downstream task outside the scope table."""
import torch
from torch import nn

from .dense import hip_autograd, hip_inference, linear_autograd, linear_forward


class MergeLayer(nn.Module):
    """Two-layer perceptron on a concatenated pair: fc2(dropout(relu(fc1([x1 | x2])))).
    `fc1` / `fc2` / `dropout` are the reference's attribute names (state_dict keys, dropout
    probability read by the training step).  Inside the HIP path the two layers are consumed
    as raw weights (tg_linear); on the operator path this forward runs the same kernels - under no_grad directly,
    under autograd through an autograd function whose backward is the library's too (tg_linear_bwd)."""

    def __init__(self, dim1: int, dim2: int, hidden_size: int, out_size: int, dropout: float = 0.):
        super().__init__()
        layers = {'fc1': nn.Linear(dim1 + dim2, hidden_size), 'fc2': nn.Linear(hidden_size, out_size)}
        for name, layer in layers.items():
            nn.init.xavier_normal_(layer.weight)  # biases keep nn.Linear's default, as in the reference
            self.add_module(name, layer)
        self.dropout = nn.Dropout(dropout)
        self.act = nn.ReLU()

    def forward(self, x1, x2):
        x = torch.cat((x1, x2), dim=-1)
        if hip_inference(x, self.dropout, self.fc1, self.fc2):  # the library's MFMA kernels (tg_linear_fwd), ReLU fused
            return linear_forward(self.fc2, linear_forward(self.fc1, x, relu=True))
        if hip_autograd(x, self.fc1, self.fc2):  # autograd / active dropout: the same kernels through an autograd function
            return linear_autograd(self.fc2, self.dropout(self.act(linear_autograd(self.fc1, x))))
        return self.fc2(self.dropout(self.act(self.fc1(x))))  # CPU tensors, widths the kernels do not take: plain torch
