"""Mirror of tiger/model/utils.py: device implementations of the two index helpers."""
import numpy as np
import torch

from .. import hip_ops


def select_latest_nids(nids, ts, n_nodes=None):
    """(unique_ids sorted, index of the latest occurrence; first index among equal
    timestamps - the torch_scatter CPU tie rule).  Accepts CPU or device tensors; CPU
    inputs are staged through the current device and returned on the CPU."""
    was_cpu = not nids.is_cuda
    if was_cpu:
        nids, ts = nids.cuda(), ts.cuda()
    u, idx = hip_ops.select_latest_nids(nids, ts, n_nodes)
    return (u.cpu(), idx.cpu()) if was_cpu else (u, idx)


def anonymized_reindex(hist_nids: np.ndarray) -> np.ndarray:
    """numpy in / numpy out like the reference (utils.py:19-27); runs on the device."""
    t = torch.from_numpy(np.ascontiguousarray(hist_nids, dtype=np.int64)).cuda()
    return hip_ops.anonymized_reindex(t).cpu().numpy().astype(hist_nids.dtype, copy=False)
