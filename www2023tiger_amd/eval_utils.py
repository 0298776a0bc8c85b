"""Mirror of tiger/eval_utils.py for the link-prediction path: `eval_edge_prediction`
(eval_utils.py:15-68) and `warmup` (eval_utils.py:102-129).  The model forward is the same
device path as in training; scores stay on the GPU until the end, where AP / AUC per window of
`mean_over_n_samples` events come from one kernel (`tg_ap_auc`) instead of sklearn round trips.
Node classification and trajectory encoding (eval_utils.py:71-99,132-183) are downstream tasks
outside the scope table."""
import math
import warnings
from typing import Optional

import torch

from ._lib import check, lib, ptr
from .hip_ops import stream_ptr
from .utils import BackgroundThreadGenerator


def ap_auc_windows(pos_pred: torch.Tensor, neg_pred: torch.Tensor, window: int = 200):
    """sklearn average_precision_score / roc_auc_score of every window of `window` events."""
    n = pos_pred.numel()
    dev = pos_pred.device
    nw = math.ceil(n / window) if n else 0
    ap = torch.zeros(nw, dtype=torch.float64, device=dev)
    auc = torch.zeros(nw, dtype=torch.float64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    pos_pred, neg_pred = pos_pred.float().contiguous(), neg_pred.float().contiguous()
    check(lib.tg_ap_auc(n, window, ptr(pos_pred), ptr(neg_pred), ptr(ap), ptr(auc), ptr(bad), stream_ptr(dev)),
          'tg_ap_auc')
    return ap, auc, bad


def _lazy_restart(model, comp_graph, ts, uptodate_nodes: set, device):
    """eval_utils.py:37-42: restart the involved nodes that are not up to date yet."""
    involved_nodes = comp_graph.np_computation_graph_nodes
    restart_nodes = set(involved_nodes.tolist()) - uptodate_nodes
    r_nids = torch.tensor(sorted(restart_nodes), dtype=torch.long, device=device)
    model.restart(r_nids, torch.full((len(r_nids),), ts.min().item(), device=device))
    uptodate_nodes.update(restart_nodes)


def eval_edge_prediction(model, dl, device: torch.device, restart_mode: bool, uptodate_nodes: Optional[set] = None,
                         mean_over_n_samples: int = 200):
    """-> (mean AP, mean AUC) over windows of `mean_over_n_samples` events.  `uptodate_nodes` is
    updated in place when given (as in the reference)."""
    model.eval()
    uptodate_nodes = set() if uptodate_nodes is None else uptodate_nodes
    pos_all, neg_all = [], []
    with torch.no_grad():
        for src_ids, dst_ids, neg_dst_ids, ts, eids, _, comp_graph in BackgroundThreadGenerator(dl):
            src_ids, dst_ids, neg_dst_ids = (x.long().to(device) for x in (src_ids, dst_ids, neg_dst_ids))
            ts, eids = ts.float().to(device), eids.long().to(device)
            comp_graph.to(device)
            if restart_mode:
                _lazy_restart(model, comp_graph, ts, uptodate_nodes, device)
            _, _, pos_scores, neg_scores, *_ = model.contrast_learning(src_ids, dst_ids, neg_dst_ids, ts, eids,
                                                                       comp_graph)
            pos_all.append(pos_scores.sigmoid())
            neg_all.append(neg_scores.sigmoid())
    model._poll_train_errors()  # the last batch's invariant word (the one-call step reads it back asynchronously)
    if not pos_all:
        return float('nan'), float('nan')
    ap, auc, bad = ap_auc_windows(torch.cat(pos_all), torch.cat(neg_all), mean_over_n_samples)
    if int(bad.item()):
        warnings.warn(f'Encounter invalid values: {int(bad.item())} non-finite predictions were dropped')
    return float(ap.mean().item()), float(auc.mean().item())


def warmup(model, dl, device: torch.device, uptodate_nodes: Optional[set] = None):
    """Only valid in restart mode: stream the loader through the model with lazy restarts."""
    model.eval()
    uptodate_nodes = set() if uptodate_nodes is None else uptodate_nodes
    with torch.no_grad():
        for src_ids, dst_ids, neg_dst_ids, ts, eids, _, comp_graph in BackgroundThreadGenerator(dl):
            src_ids, dst_ids, neg_dst_ids = (x.long().to(device) for x in (src_ids, dst_ids, neg_dst_ids))
            ts, eids = ts.float().to(device), eids.long().to(device)
            comp_graph.to(device)
            _lazy_restart(model, comp_graph, ts, uptodate_nodes, device)
            model.contrast_learning(src_ids, dst_ids, neg_dst_ids, ts, eids, comp_graph)
    model._poll_train_errors()
    return uptodate_nodes
