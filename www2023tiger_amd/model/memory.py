"""Mirror of tiger/model/memory.py: `Memory` and `MessageStoreNoGradLastOnly`.

The dict-of-lists stores (MessageStore, MessageStoreNoGrad) are unreachable in the
reference (init_utils.py:166 hard-codes msg_last_only=True) and are not built.
State lives in torch buffers with the reference's names so checkpoints interchange;
the Python set `nodes_with_messages` becomes a device bitmap (`has_msg_bits`).
"""
import copy
from typing import Optional, Tuple, Union

import numpy as np
import torch
from torch import Tensor, nn

from .. import hip_ops
from .utils import select_latest_nids


class Memory(nn.Module):
    def __init__(self, n, dim):
        super().__init__()
        self.n = n
        self.dim = dim
        self.register_buffer('vals', torch.zeros(n, dim), persistent=True)
        self.register_buffer('update_ts', torch.zeros(n), persistent=True)
        self.register_buffer('active_mask', torch.zeros(n).bool(), persistent=True)
        self._version_ = 0  # bumped by every mutating method (the model's eager-update table watches it)

    def clone(self):
        """memory.py:21-25 - like the reference, active_mask is not carried over."""
        other = Memory(self.n, self.dim).to(self.device)
        other.vals.copy_(self.vals)
        other.update_ts.copy_(self.update_ts)
        return other

    @property
    def device(self):
        return self.vals.device

    def clear(self):
        self._version_ += 1
        self.vals.zero_()
        self.update_ts.zero_()
        self.active_mask.zero_()

    def get(self, ids: Tensor) -> Tuple[Tensor, Tensor]:
        return hip_ops.gather_rows(self.vals, ids, self.update_ts)

    def set(self, ids: Tensor, vals: Tensor, ts: Tensor, skip_check=False):
        if len(ids) == 0:
            return
        self._version_ += 1
        err = None
        if not skip_check:
            if len(ids) != len(torch.unique(ids)):
                raise ValueError('Duplicate node ids are not allowed.')
            err = hip_ops.new_err(self.device)
        hip_ops.memory_scatter(self.vals, self.update_ts, self.active_mask, ids, vals.detach(), ts,
                               check_past=not skip_check, err=err)
        if err is not None:
            hip_ops.raise_if_err(err)


class MessageStoreNoGradLastOnly(nn.Module):
    """Last-message mailbox: one raw message row [own | other | edge | time] per node."""

    def __init__(self, n, dim):
        super().__init__()
        self.n = n
        self.dim = dim
        self.register_buffer('node_msg_vals', torch.zeros((n, dim)).float(), persistent=False)
        self.register_buffer('node_msg_ts', torch.zeros(n).float(), persistent=False)
        self.register_buffer('has_msg_bits', torch.zeros(hip_ops.bitmap_words(n), dtype=torch.int64), persistent=False)
        self._version_ = 0  # bumped by every mutating method

    @property
    def node_messages(self):
        return (self.node_msg_vals, self.node_msg_ts)

    @property
    def device(self):
        return self.node_msg_vals.device

    def clone(self):
        return copy.deepcopy(self)

    # -- set view of the bitmap (host sync; diagnostic / compatibility only) --------
    def _has_msg_numpy(self) -> np.ndarray:
        words = self.has_msg_bits.cpu().numpy().view(np.uint64)
        bits = np.unpackbits(words.view(np.uint8), bitorder='little')[:self.n]
        return np.nonzero(bits)[0]

    @property
    def nodes_with_messages(self) -> set:
        return set(self._has_msg_numpy().tolist())

    def get_outdated_node_ids(self, node_ids: Union[Tensor, np.ndarray, None]) -> Tensor:
        """memory.py:108-126: ids (a subset of node_ids) that hold an unconsumed message,
        as a CPU LongTensor (sorted here; the reference returns Python-set order)."""
        has = self._has_msg_numpy()
        if node_ids is not None:
            ids = node_ids if isinstance(node_ids, np.ndarray) else node_ids.cpu().numpy()
            has = np.intersect1d(has, ids)
        return torch.from_numpy(has.astype(np.int64))

    def clear(self, nids: Optional[Tensor] = None):
        """memory.py:128-138 (the reference's zero-fill of rows is a no-op on an indexed copy;
        only membership changes)."""
        self._version_ += 1
        if nids is None:
            self.has_msg_bits.zero_()
            return
        if len(nids) == 0:
            return
        nids = nids.to(self.device).long()
        mask = torch.zeros_like(self.has_msg_bits)
        hip_ops.bitmap_mark(nids, mask, self.n)
        self.has_msg_bits &= ~mask

    def store_events(self, src_ids, dst_ids, src_prev_ts, dst_prev_ts, src_vals, dst_vals, eids, ts, emb_getter,
                     time_encoder):
        """memory.py:77-106, composed from the standalone ops.  TIGE.store_events uses the
        fused tg_store_events kernel instead; this form exists for API compatibility."""
        self._version_ += 1
        pos = torch.cat([src_ids, dst_ids])
        if bool((self._bits_of(pos)).any()):
            raise ValueError('Node has unused messages.')
        sv = src_vals + emb_getter.get_node_embeddings(src_ids)
        dv = dst_vals + emb_getter.get_node_embeddings(dst_ids)
        ev = emb_getter.get_edge_embeddings(eids)
        full = torch.cat([torch.cat([sv, dv, ev, time_encoder(ts - src_prev_ts)], 1),
                          torch.cat([dv, sv, ev, time_encoder(ts - dst_prev_ts)], 1)], 0)
        ts2 = ts.repeat(2)
        ids, index = select_latest_nids(pos, ts2, self.n)
        hip_ops.memory_scatter(self.node_msg_vals, self.node_msg_ts, None, ids, full, ts2, src_index=index)
        hip_ops.bitmap_mark(ids, self.has_msg_bits, self.n)

    def has_msg_mask(self) -> Tensor:
        """bool[n] on the device: node holds an unconsumed message (the bitmap, one flag per node)"""
        return self._bits_of(torch.arange(self.n, device=self.device)).bool()

    def _bits_of(self, ids: Tensor) -> Tensor:
        return (self.has_msg_bits[ids >> 6] >> (ids & 63)) & 1
