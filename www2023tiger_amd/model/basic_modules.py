"""Mirror of tiger/model/basic_modules.py (MergeLayer only; the node-classification MLP is out of scope)."""
import torch
from torch import nn


class MergeLayer(nn.Module):
    """fc2(dropout(relu(fc1(cat[x1, x2])))) with Xavier-normal weights.  Inside the HIP
    path the two Linear layers are consumed as raw weights (tg_linear); this forward is
    the plain torch form used by the score head (STEP 7, outside the timed path)."""

    def __init__(self, dim1, dim2, hidden_size, out_size, dropout=0.):
        super().__init__()
        self.fc1 = nn.Linear(dim1 + dim2, hidden_size)
        self.fc2 = nn.Linear(hidden_size, out_size)
        self.dropout = nn.Dropout(dropout)
        self.act = nn.ReLU()
        nn.init.xavier_normal_(self.fc1.weight)
        nn.init.xavier_normal_(self.fc2.weight)

    def forward(self, x1, x2):
        return self.fc2(self.dropout(self.act(self.fc1(torch.cat([x1, x2], dim=-1)))))
