"""Mirror of the reference's init_utils.py:64-167 (`init_data`, `init_model`): the objects its
training scripts build, on this package's classes.  The argument parser / CLI is out of scope."""
import math

import torch

from .data.data_loader import BatchLoader, ChunkSampler, GraphCollator, load_jodie_data
from .data.graph import Graph
from .model.feature_getter import NumericalFeature
from .model.restarters import SeqRestarter, StaticRestarter
from .model.tiger import TIGER


def init_data(data, root, seed, rank=None, world_size=None, *, num_workers=0, bs, warmup_steps, subset, strategy,
              n_layers, n_neighbors, restarter_type, hist_len, device=None):
    """-> (basic_data, (train_graph, full_graph), (train_dl, offline_dl, val_dl, ind_val_dl, test_dl,
    ind_test_dl, val_warmup_dl, test_warmup_dl)).  Collation runs on `device` (default cuda:0), so
    loaders use no worker processes."""
    if num_workers:
        raise ValueError('collation uses the GPU: num_workers must be 0')
    device = torch.device('cuda', 0) if device is None else torch.device(device)
    basic = load_jodie_data(data, train_seed=seed, root=root)
    nfeats, efeats, full_data, train_data, val_data, test_data, ind_val_data, ind_test_data = basic
    offline_data = None
    if subset < 1.0:
        cut = math.ceil(len(train_data) * subset)
        offline_data = train_data.get_subset(cut, len(train_data))
        train_data = train_data.get_subset(0, cut)
    # both graphs over the id space of the FULL data: the model's tables (n_nodes = full_graph.num_node,
    # init_utils.py:141) and the collator's bitmaps are sized by it, while the training split usually lacks the
    # highest ids (items first seen after the validation time, the hidden 10 % of the nodes).  Rows of nodes without
    # training events are empty, so sampling on the training graph is what the reference's smaller graph gives
    max_id = int(max(full_data.src.max(), full_data.dst.max()))
    train_graph = Graph.from_data(train_data, strategy=strategy, seed=seed, max_node_id=max_id, device=device)
    full_graph = Graph.from_data(full_data, strategy=strategy, seed=seed, max_node_id=max_id, device=device)
    mk_coll = lambda g: GraphCollator(g, n_neighbors, n_layers, restarter=restarter_type, hist_len=hist_len)
    train_coll, eval_coll = mk_coll(train_graph), mk_coll(full_graph)
    # same iteration protocol as the reference's DataLoaders, batches sliced natively (no per-event Python)
    loader = lambda ds, coll, **kw: BatchLoader(ds, bs, coll, **kw)
    if world_size is not None:  # the reference's DDP recipe: one time chunk per rank
        sampler = ChunkSampler(len(train_data), rank=rank, world_size=world_size, bs=bs, seed=seed)
        train_dl, offline_dl = loader(train_data, train_coll, sampler=sampler), None
    else:
        train_dl = loader(train_data, train_coll)
        offline_dl = loader(offline_data, eval_coll) if offline_data is not None else None
    val_warm = test_warm = None
    if warmup_steps > 0:
        if warmup_steps > len(train_data) or warmup_steps > len(val_data):
            raise ValueError('Too many warmup steps!')
        val_warm = loader(train_data.get_subset(len(train_data) - warmup_steps, len(train_data)), train_coll)
        test_warm = loader(val_data.get_subset(len(val_data) - warmup_steps, len(val_data)), eval_coll)
    dls = (train_dl, offline_dl, loader(val_data, eval_coll), loader(ind_val_data, eval_coll),
           loader(test_data, eval_coll), loader(ind_test_data, eval_coll), val_warm, test_warm)
    basic = (nfeats, efeats, full_data, train_data, val_data, test_data, ind_val_data, ind_test_data)
    return basic, (train_graph, full_graph), dls


def init_model(nfeats, efeats, train_graph, full_graph, full_data, device, *, feature_as_buffer=True, dim, n_layers,
               n_heads, n_neighbors, hit_type, dropout, restarter_type, hist_len, msg_src, upd_src, msg_tsfm_type,
               mem_update_type):
    to_t = lambda a: None if a is None else torch.from_numpy(a).float()
    nfeats, efeats = to_t(nfeats), to_t(efeats)
    for t in (nfeats, efeats):  # the first table present fixes the width when --dim is not given
        if t is not None and dim is None:
            dim = t.shape[1]
    # --no_feat_buffer: the tables stay in pinned host memory and the kernels read them in place
    getter = NumericalFeature(nfeats, efeats, dim=dim, register_buffer=feature_as_buffer, device=device)
    getter.n_nodes, getter.n_edges = full_graph.num_node, len(full_data)
    if restarter_type == 'seq':
        restarter = SeqRestarter(raw_feat_getter=getter, graph=train_graph, hist_len=hist_len, n_head=n_heads,
                                 dropout=dropout)
    elif restarter_type == 'static':
        restarter = StaticRestarter(raw_feat_getter=getter, graph=train_graph)
    else:
        raise NotImplementedError(restarter_type)
    model = TIGER(raw_feat_getter=getter, graph=train_graph, restarter=restarter, n_neighbors=n_neighbors,
                  hit_type=hit_type, n_layers=n_layers, n_head=n_heads, dropout=dropout, msg_src=msg_src,
                  upd_src=upd_src, msg_tsfm_type=msg_tsfm_type, mem_update_type=mem_update_type, tgn_mode=True,
                  msg_last_only=True)
    return model.to(device)
