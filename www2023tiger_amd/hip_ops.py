"""Thin torch-tensor wrappers around the C ABI (one function per entry point).

Tensors are plumbing: they own device memory and name the stream; all work
happens in libtiger_hip.so.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, lib, ptr


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _i64(t: Tensor) -> Tensor:
    return t.contiguous() if t.dtype == torch.int64 else t.long().contiguous()


def new_err(device) -> Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


def raise_if_err(err: Tensor):
    """Reads the invariant word back (one host sync) and raises the reference's ValueError."""
    _lib.raise_invariants(int(err.item()) & 0xFFFFFFFF)


def bitmap_words(n_nodes: int) -> int:
    return int(lib.tg_bitmap_words(n_nodes))


def new_bitmap(n_nodes: int, device) -> Tensor:
    return torch.zeros(bitmap_words(n_nodes), dtype=torch.int64, device=device)


def bitmap_mark(ids: Tensor, bitmap: Tensor, n_nodes: int):
    ids = _i64(ids)
    check(lib.tg_bitmap_mark(ids.numel(), ptr(ids), ptr(bitmap), n_nodes, stream_ptr(ids.device)), 'tg_bitmap_mark')


def new_flags(n_nodes: int, device) -> Tensor:
    """zeroed byte flags, one per node (padded to a multiple of 64)"""
    return torch.zeros(int(lib.tg_flag_bytes(n_nodes)), dtype=torch.uint8, device=device)


def flags_mark(ids: Tensor, flags: Tensor, n_nodes: int):
    ids = _i64(ids)
    check(lib.tg_flags_mark(ids.numel(), ptr(ids), ptr(flags), n_nodes, stream_ptr(ids.device)), 'tg_flags_mark')


def unique_compact(bitmap: Optional[Tensor], n_nodes: int, cap: int, and_bitmap: Optional[Tensor] = None,
                   flags: Optional[Tensor] = None):
    """-> dict(bitmap, rank, ids, count [, and_rank, and_ids, and_pos, and_count]); lists have
    capacity `cap`.  Either `bitmap` (input) or `flags` (packed into a fresh bitmap) is given."""
    dev = (bitmap if bitmap is not None else flags).device
    if bitmap is None:
        bitmap = torch.empty(bitmap_words(n_nodes), dtype=torch.int64, device=dev)
    W = bitmap_words(n_nodes)
    out = dict(bitmap=bitmap, rank=torch.empty(W + 1, dtype=torch.int32, device=dev),
               ids=torch.empty(cap, dtype=torch.int64, device=dev),
               count=torch.zeros(1, dtype=torch.int32, device=dev))
    a = [None] * 4
    if and_bitmap is not None:
        out.update(and_rank=torch.empty(W + 1, dtype=torch.int32, device=dev),
                   and_ids=torch.empty(cap, dtype=torch.int64, device=dev),
                   and_pos=torch.empty(cap, dtype=torch.int32, device=dev),
                   and_count=torch.zeros(1, dtype=torch.int32, device=dev))
        a = [out['and_rank'], out['and_ids'], out['and_pos'], out['and_count']]
    nbytes = int(lib.tg_unique_compact_workspace_bytes(n_nodes))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    check(lib.tg_unique_compact(ptr(flags), ptr(bitmap), n_nodes, ptr(out['rank']), ptr(out['ids']), ptr(out['count']), cap,
                                ptr(and_bitmap), ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(ws), ws.numel(),
                                stream_ptr(dev)), 'tg_unique_compact')
    return out


def select_latest_nids(nids: Tensor, ts: Tensor, n_nodes: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """tiger/model/utils.py:10-16 on device: (sorted unique ids, position of the latest
    occurrence, first position among equal timestamps)."""
    nids = _i64(nids)
    dev = nids.device
    n = nids.numel()
    if n == 0:
        return nids.new_empty(0), nids.new_empty(0)
    if ts.dtype not in (torch.float32, torch.float64):
        ts = ts.double()
    ts = ts.contiguous()
    if n_nodes is None:
        n_nodes = int(nids.max().item()) + 1
    uniq = torch.empty(n, dtype=torch.int64, device=dev)
    index = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    nbytes = int(lib.tg_select_latest_workspace_bytes(n, n_nodes))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib.tg_select_latest(n, ptr(nids), ptr(ts), 1 if ts.dtype == torch.float64 else 0, n_nodes, ptr(uniq),
                               ptr(index), ptr(count), ptr(ws), nbytes, stream_ptr(dev)), 'tg_select_latest')
    p = int(count.item())
    return uniq[:p], index[:p]


def anonymized_reindex(hist_nids: Tensor) -> Tensor:
    """tiger/model/utils.py:19-27 on an [n, H] id matrix (H <= 64)."""
    hist_nids = _i64(hist_nids)
    out = torch.empty_like(hist_nids)
    n, H = hist_nids.shape
    check(lib.tg_anonymized_reindex(n, H, ptr(hist_nids), ptr(out), stream_ptr(hist_nids.device)),
          'tg_anonymized_reindex')
    return out


def hits(center: Tensor, nbr: Tensor) -> Tensor:
    """data_loader.py:61-67: (center[:, None] == nbr) as float32."""
    center, nbr = _i64(center), _i64(nbr)
    B, K = nbr.shape
    out = torch.empty(B, K, dtype=torch.float32, device=nbr.device)
    check(lib.tg_hits(B, K, ptr(center), ptr(nbr), ptr(out), stream_ptr(nbr.device)), 'tg_hits')
    return out


def time_encode(ts: Tensor, freq: Tensor, phase: Tensor) -> Tensor:
    """time_encoding.py:24-26."""
    flat = ts.contiguous().float().reshape(-1)
    d = freq.numel()
    out = torch.empty(flat.numel(), d, dtype=torch.float32, device=ts.device)
    check(lib.tg_time_encode(flat.numel(), ptr(flat), d, ptr(freq), ptr(phase), ptr(out), stream_ptr(ts.device)),
          'tg_time_encode')
    return out.reshape(*ts.shape, d)


def gather_rows(table: Tensor, ids: Tensor, ts_table: Optional[Tensor] = None):
    ids_f = _i64(ids).reshape(-1)
    width = table.shape[1]
    dev = ids_f.device if table.device.type == 'cpu' else table.device  # a pinned host table is read from the ids' GPU
    out = torch.empty(ids_f.numel(), width, dtype=torch.float32, device=dev)
    ts_out = torch.empty(ids_f.numel(), dtype=torch.float32, device=dev) if ts_table is not None else None
    check(lib.tg_gather_rows(ids_f.numel(), ptr(ids_f), width, ptr(table), ptr(out), ptr(ts_table), ptr(ts_out),
                             stream_ptr(dev)), 'tg_gather_rows')
    out = out.reshape(*ids.shape, width)
    if ts_table is None:
        return out
    return out, ts_out.reshape(ids.shape)


def memory_scatter(table: Tensor, ts_table: Tensor, active: Optional[Tensor], ids: Tensor, vals: Tensor, ts: Tensor,
                   src_index: Optional[Tensor] = None, check_past: bool = False, err: Optional[Tensor] = None):
    ids = _i64(ids)
    vals = vals.contiguous().float()
    ts = ts.contiguous().float()
    if src_index is not None:
        src_index = _i64(src_index)
    if check_past and err is None:
        raise ValueError('check_past needs an err word')
    check(lib.tg_memory_scatter(ids.numel(), None, ptr(ids), ptr(src_index), table.shape[1], ptr(vals), ptr(ts),
                                ptr(table), ptr(ts_table), ptr(active), 1 if check_past else 0, ptr(err),
                                stream_ptr(table.device)), 'tg_memory_scatter')
