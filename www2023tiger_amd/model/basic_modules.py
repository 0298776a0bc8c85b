"""MergeLayer (reference: tiger/model/basic_modules.py:5-19); the node-classification MLP of that
file belongs to a downstream task outside the scope table."""
import torch
from torch import nn


class MergeLayer(nn.Module):
    """Two-layer perceptron on a concatenated pair: fc2(dropout(relu(fc1([x1 | x2])))).
    `fc1` / `fc2` / `dropout` are the reference's attribute names (state_dict keys, dropout
    probability read by the training step).  Inside the HIP path the two layers are consumed
    as raw weights (tg_linear); this forward is the plain torch form of the operator path."""

    def __init__(self, dim1: int, dim2: int, hidden_size: int, out_size: int, dropout: float = 0.):
        super().__init__()
        layers = {'fc1': nn.Linear(dim1 + dim2, hidden_size), 'fc2': nn.Linear(hidden_size, out_size)}
        for name, layer in layers.items():
            nn.init.xavier_normal_(layer.weight)  # biases keep nn.Linear's default, as in the reference
            self.add_module(name, layer)
        self.dropout = nn.Dropout(dropout)
        self.act = nn.ReLU()

    def forward(self, x1, x2):
        hidden = self.act(self.fc1(torch.cat((x1, x2), dim=-1)))
        return self.fc2(self.dropout(hidden))
