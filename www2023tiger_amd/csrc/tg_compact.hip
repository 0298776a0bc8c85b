// Sorted-unique compaction on node bitmaps (a5, a8, a10, a12).
// Replaces np.sort(list(set)) (data_loader.py:109-121), the dense local_index
// (data_classes.py:163-165), the Python-set intersection of get_outdated_node_ids
// (memory.py:108-126) and torch.unique + scatter_max (tiger/model/utils.py:10-16).
// A bitmap over node ids is its own sorted order, so no sort is needed: ranks are
// prefix popcounts.
#include "tg_common.h"

namespace tg {

constexpr int CB = 256;  // threads per block = bitmap words per block

__global__ void k_mark(int64_t n, const int64_t* __restrict__ ids, uint64_t* __restrict__ bm, int64_t n_nodes) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = ids[i];
    if (id < 0 || id >= n_nodes) continue;
    const uint64_t bit = 1ull << (id & 63);
    uint64_t* w = bm + (id >> 6);
    if ((__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0)
      atomicOr((unsigned long long*)w, bit);
  }
}

// byte flags: plain stores, no atomics (all writers store the same value)
__global__ void k_mark_flags(int64_t n, const int64_t* __restrict__ ids, uint8_t* __restrict__ flags, int64_t n_nodes) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = ids[i];
    if (id >= 0 && id < n_nodes) flags[id] = 1;
  }
}

// exclusive scan of one value per thread over a 256-thread block; returns the block total in *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t* total) {
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < TG_WAVE; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, TG_WAVE);
    if (lane >= o) inc += t;
  }
  if (lane == TG_WAVE - 1) s_wave[wv] = inc;
  __syncthreads();
  uint32_t base = 0;
  uint32_t tot = 0;
#pragma unroll
  for (int i = 0; i < CB / TG_WAVE; ++i) {
    const uint32_t s = s_wave[i];
    if (i < wv) base += s;
    tot += s;
  }
  *total = tot;
  __syncthreads();
  return base + inc - v;
}

// 8 flag bytes (each 0/1) -> 8 bits
__device__ __forceinline__ uint64_t pack8(uint64_t x) { return ((x & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56; }

// phase A: one thread per bitmap word.  Optionally packs 64 flag bytes into the word first
// (the bitmap is then an OUTPUT).  Writes block-relative exclusive ranks and block totals.
__global__ void __launch_bounds__(CB) k_bm_pack_scan(const uint8_t* __restrict__ flags, uint64_t* __restrict__ bm,
                                                     const uint64_t* __restrict__ hm, int64_t W,
                                                     uint32_t* __restrict__ rank1, uint32_t* __restrict__ rank2,
                                                     uint32_t* __restrict__ blk1, uint32_t* __restrict__ blk2,
                                                     int32_t* __restrict__ count1, int32_t* __restrict__ count2) {
  __shared__ uint32_t s_w[2][CB / TG_WAVE];
  const int64_t w = (int64_t)blockIdx.x * CB + threadIdx.x;
  uint64_t a = 0, b = 0;
  if (w < W) {
    if (flags) {
      const uint4* f = reinterpret_cast<const uint4*>(flags + w * 64);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 v = f[q];
        const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
        a |= (pack8(lo) | (pack8(hi) << 8)) << (16 * q);
      }
      bm[w] = a;
    } else {
      a = bm[w];
    }
    if (hm) b = a & hm[w];
  }
  uint32_t t1, t2;
  const uint32_t r1 = block_excl_scan((uint32_t)__popcll(a), s_w[0], &t1);
  const uint32_t r2 = block_excl_scan((uint32_t)__popcll(b), s_w[1], &t2);
  if (w < W) {
    rank1[w] = r1;
    if (rank2) rank2[w] = r2;
  }
  if (threadIdx.x == 0) {
    blk1[blockIdx.x] = t1;
    blk2[blockIdx.x] = t2;
    if (gridDim.x == 1) {
      if (count1) *count1 = (int32_t)t1;
      if (count2) *count2 = (int32_t)t2;
    }
  }
}

// Small graphs (<= 1024 words = 65536 nodes): pack + scan + emit in ONE 1024-thread block.
constexpr int SB = 1024;
__global__ void __launch_bounds__(SB) k_bm_small(const uint8_t* __restrict__ flags, uint64_t* __restrict__ bm,
                                                 const uint64_t* __restrict__ hm, int W, uint32_t* __restrict__ rank1,
                                                 int64_t* __restrict__ ids1, int32_t* __restrict__ count1, int64_t cap,
                                                 uint32_t* __restrict__ rank2, int64_t* __restrict__ ids2,
                                                 int32_t* __restrict__ pos2, int32_t* __restrict__ count2) {
  constexpr int NWV = SB / TG_WAVE;
  __shared__ uint32_t s_w1[NWV], s_w2[NWV];
  __shared__ uint64_t s_a[SB], s_b[SB];
  __shared__ uint32_t s_r1[SB], s_r2[SB];
  const int w = threadIdx.x, lane = lane_id(), wv = threadIdx.x >> 6;
  uint64_t a = 0, b = 0;
  if (w < W) {
    if (flags) {
      const uint4* f = reinterpret_cast<const uint4*>(flags + (int64_t)w * 64);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 v = f[q];
        const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
        a |= (pack8(lo) | (pack8(hi) << 8)) << (16 * q);
      }
      bm[w] = a;
    } else {
      a = bm[w];
    }
    if (hm) b = a & hm[w];
  }
  // block-wide exclusive scan of both popcounts: wave scan, then the 16 wave totals
  uint32_t i1 = (uint32_t)__popcll(a), i2 = (uint32_t)__popcll(b);
  const uint32_t c1 = i1, c2 = i2;
#pragma unroll
  for (int o = 1; o < TG_WAVE; o <<= 1) {
    const uint32_t t1 = __shfl_up(i1, o, TG_WAVE), t2 = __shfl_up(i2, o, TG_WAVE);
    if (lane >= o) {
      i1 += t1;
      i2 += t2;
    }
  }
  if (lane == TG_WAVE - 1) {
    s_w1[wv] = i1;
    s_w2[wv] = i2;
  }
  __syncthreads();
  uint32_t base1 = 0, base2 = 0, t1 = 0, t2 = 0;
#pragma unroll
  for (int i = 0; i < NWV; ++i) {
    const uint32_t x1 = s_w1[i], x2 = s_w2[i];
    if (i < wv) {
      base1 += x1;
      base2 += x2;
    }
    t1 += x1;
    t2 += x2;
  }
  const uint32_t r1 = base1 + i1 - c1, r2 = base2 + i2 - c2;
  s_a[w] = a;
  s_b[w] = b;
  s_r1[w] = r1;
  s_r2[w] = r2;
  if (w < W) {
    rank1[w] = r1;
    if (rank2) rank2[w] = r2;
  }
  if (w == 0) {
    rank1[W] = t1;
    if (rank2) rank2[W] = t2;
    if (count1) *count1 = (int32_t)t1;
    if (count2) *count2 = (int32_t)t2;
  }
  __syncthreads();
  const uint64_t below = (1ull << lane) - 1ull;
  for (int ww = wv; ww < W; ww += NWV) {  // one wavefront per word, one lane per bit
    const uint64_t wa = s_a[ww], wb = s_b[ww];
    const uint32_t mine = s_r1[ww] + (uint32_t)__popcll(wa & below);
    if (((wa >> lane) & 1ull) && ids1 && (int64_t)mine < cap) ids1[mine] = (int64_t)ww * 64 + lane;
    if (rank2) {
      const uint32_t m2 = s_r2[ww] + (uint32_t)__popcll(wb & below);
      if (((wb >> lane) & 1ull) && (int64_t)m2 < cap) {
        if (ids2) ids2[m2] = (int64_t)ww * 64 + lane;
        if (pos2) pos2[m2] = (int32_t)mine;
      }
    }
  }
}

// phase B (only when more than one block): block totals -> exclusive block offsets + counts
__global__ void __launch_bounds__(CB) k_bm_scan_blocks(uint32_t* __restrict__ blk1, uint32_t* __restrict__ blk2, int nblk,
                                                       int32_t* __restrict__ count1, int32_t* __restrict__ count2) {
  __shared__ uint32_t s_w[2][CB / TG_WAVE];
  uint32_t run1 = 0, run2 = 0;
  for (int base = 0; base < nblk; base += CB) {
    const int i = base + threadIdx.x;
    const uint32_t v1 = i < nblk ? blk1[i] : 0, v2 = i < nblk ? blk2[i] : 0;
    uint32_t t1, t2;
    const uint32_t e1 = block_excl_scan(v1, s_w[0], &t1);
    const uint32_t e2 = block_excl_scan(v2, s_w[1], &t2);
    if (i < nblk) {
      blk1[i] = run1 + e1;
      blk2[i] = run2 + e2;
    }
    run1 += t1;
    run2 += t2;
  }
  if (threadIdx.x == 0) {
    if (count1) *count1 = (int32_t)run1;
    if (count2) *count2 = (int32_t)run2;
  }
}

// phase C: one wavefront per word, one lane per bit: coalesced emission of the id lists;
// also turns the block-relative ranks into global ones.
__global__ void __launch_bounds__(CB) k_bm_emit(const uint64_t* __restrict__ bm, const uint64_t* __restrict__ hm, int64_t W,
                                                int multi, const uint32_t* __restrict__ blk1,
                                                const uint32_t* __restrict__ blk2, uint32_t* __restrict__ rank1,
                                                int64_t* __restrict__ ids1, int64_t cap, uint32_t* __restrict__ rank2,
                                                int64_t* __restrict__ ids2, int32_t* __restrict__ pos2,
                                                const int32_t* __restrict__ count1, const int32_t* __restrict__ count2) {
  const int lane = lane_id();
  const uint64_t below = (1ull << lane) - 1ull;
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < W; w += (int64_t)gridDim.x * 4) {
    const int64_t blk = w / CB;
    const uint64_t a = bm[w];
    const uint32_t r1 = rank1[w] + (multi ? blk1[blk] : 0u);
    const uint32_t mine = r1 + (uint32_t)__popcll(a & below);
    const bool set1 = (a >> lane) & 1ull;
    if (set1 && ids1 && (int64_t)mine < cap) ids1[mine] = w * 64 + lane;
    uint32_t r2 = 0;
    if (rank2) {
      const uint64_t b = hm ? (a & hm[w]) : 0ull;
      r2 = rank2[w] + (multi ? blk2[blk] : 0u);
      const uint32_t m2 = r2 + (uint32_t)__popcll(b & below);
      if (((b >> lane) & 1ull) && (int64_t)m2 < cap) {
        if (ids2) ids2[m2] = w * 64 + lane;
        if (pos2) pos2[m2] = (int32_t)mine;
      }
    }
    if (multi && lane == 0) {
      rank1[w] = r1;
      if (rank2) rank2[w] = r2;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // sentinel entries rank[W] = totals
    rank1[W] = (uint32_t)*count1;
    if (rank2) rank2[W] = (uint32_t)*count2;
  }
}

template <typename T>
__global__ void k_sel_max(int64_t n, const int64_t* __restrict__ nids, const T* __restrict__ ts,
                          const uint64_t* __restrict__ bm, const uint32_t* __restrict__ rank,
                          unsigned long long* __restrict__ best) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    atomicMax(best + bm_rank(bm, rank, nids[i]), (unsigned long long)orderable(ts[i]));
}

template <typename T>
__global__ void k_sel_min(int64_t n, const int64_t* __restrict__ nids, const T* __restrict__ ts,
                          const uint64_t* __restrict__ bm, const uint32_t* __restrict__ rank,
                          const unsigned long long* __restrict__ best, unsigned int* __restrict__ best_idx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = bm_rank(bm, rank, nids[i]);
    // (the smallest index wins a tie: kept as the largest complement, so that the slot can start from zero like `best`)
    if ((unsigned long long)orderable(ts[i]) == best[r]) atomicMax(best_idx + r, ~(unsigned int)i);
  }
}

__global__ void k_sel_out(const int32_t* __restrict__ count, int64_t cap, const unsigned int* __restrict__ best_idx,
                          int64_t* __restrict__ out_index) {
  const int64_t n = *count;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n && i < cap; i += (int64_t)gridDim.x * blockDim.x)
    out_index[i] = (int64_t)(~best_idx[i]);
}

static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// scratch of the scan itself: two uint32 block-total arrays
static size_t scan_ws_bytes(int64_t W) { return 2 * align16((size_t)cdiv(W, CB) * sizeof(uint32_t)) + 32; }

// flags (nullable): u8[W*64] marks to pack into `bm` first (bm is then written, not read)
int unique_compact_launch(const uint8_t* flags, uint64_t* bm, int64_t n_nodes, uint32_t* rank, int64_t* ids,
                          int32_t* count, int64_t cap, const uint64_t* hm, uint32_t* rank2, int64_t* ids2,
                          int32_t* pos2, int32_t* count2, void* ws, size_t ws_bytes, hipStream_t st) {
  const int64_t W = (n_nodes + 63) / 64;
  const int nblk = (int)cdiv(W, CB);
  if (ws_bytes < scan_ws_bytes(W) || !ws) return TG_EWORKSPACE;
  uint32_t* blk1 = (uint32_t*)ws;
  uint32_t* blk2 = (uint32_t*)((char*)ws + align16((size_t)nblk * sizeof(uint32_t)));
  int32_t* spare = (int32_t*)((char*)ws + 2 * align16((size_t)nblk * sizeof(uint32_t)));
  if (!count) count = spare;       // the emit kernel needs the totals for the sentinel
  if (!count2) count2 = spare + 1;
  if (W <= SB) {
    hipLaunchKernelGGL(k_bm_small, dim3(1), dim3(SB), 0, st, flags, bm, hm, (int)W, rank, ids, count, cap, rank2, ids2,
                       pos2, count2);
    return check_launch("tg_unique_compact");
  }
  hipLaunchKernelGGL(k_bm_pack_scan, dim3(nblk), dim3(CB), 0, st, flags, bm, hm, W, rank, rank2, blk1, blk2, count,
                     count2);
  if (nblk > 1) hipLaunchKernelGGL(k_bm_scan_blocks, dim3(1), dim3(CB), 0, st, blk1, blk2, nblk, count, count2);
  hipLaunchKernelGGL(k_bm_emit, dim3(flat_grid(W, 4)), dim3(CB), 0, st, (const uint64_t*)bm, hm, W, nblk > 1 ? 1 : 0,
                     (const uint32_t*)blk1, (const uint32_t*)blk2, rank, ids, cap, rank2, ids2, pos2,
                     (const int32_t*)count, (const int32_t*)count2);
  return check_launch("tg_unique_compact");
}

}  // namespace tg

using namespace tg;

extern "C" int64_t tg_bitmap_words(int64_t n_nodes) { return (n_nodes + 63) / 64; }

extern "C" int tg_bitmap_mark(int64_t n, const int64_t* ids, uint64_t* bitmap, int64_t n_nodes, void* stream) {
  if (n < 0 || n_nodes <= 0 || !bitmap) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!ids) return TG_EINVAL;
  hipLaunchKernelGGL(k_mark, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), n, ids, bitmap, n_nodes);
  return check_launch("tg_bitmap_mark");
}

extern "C" size_t tg_unique_compact_workspace_bytes(int64_t n_nodes) { return scan_ws_bytes((n_nodes + 63) / 64); }

extern "C" int tg_unique_compact(const uint8_t* flags, uint64_t* bitmap, int64_t n_nodes, uint32_t* rank,
                                 int64_t* out_ids, int32_t* out_count, int64_t cap, const uint64_t* and_bitmap,
                                 uint32_t* and_rank, int64_t* and_ids, int32_t* and_pos, int32_t* and_count, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (!bitmap || n_nodes <= 0 || !rank || cap < 0) return TG_EINVAL;
  return unique_compact_launch(flags, bitmap, n_nodes, rank, out_ids, out_count, cap, and_bitmap, and_rank, and_ids,
                               and_pos, and_count, ws, ws_bytes, as_stream(stream));
}

extern "C" int64_t tg_flag_bytes(int64_t n_nodes) { return ((n_nodes + 63) / 64) * 64; }

extern "C" int tg_flags_mark(int64_t n, const int64_t* ids, uint8_t* flags, int64_t n_nodes, void* stream) {
  if (n < 0 || n_nodes <= 0 || !flags) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!ids) return TG_EINVAL;
  hipLaunchKernelGGL(k_mark_flags, dim3(flat_grid(n, 256)), dim3(256), 0, as_stream(stream), n, ids, flags, n_nodes);
  return check_launch("tg_flags_mark");
}

// workspace layout of tg_select_latest: flags | bitmap | rank | best (u64[n]) | best_idx (u32[n]) | scan scratch
extern "C" size_t tg_select_latest_workspace_bytes(int64_t n, int64_t n_nodes) {
  const int64_t W = (n_nodes + 63) / 64;
  return align16((size_t)W * 64) + align16((size_t)W * 8) + align16((size_t)(W + 1) * 4) + align16((size_t)n * 8) +
         align16((size_t)n * 4) + scan_ws_bytes(W);
}

extern "C" int tg_select_latest(int64_t n, const int64_t* nids, const void* ts, int32_t ts_is_f64, int64_t n_nodes,
                                int64_t* out_unique, int64_t* out_index, int32_t* out_count, void* ws, size_t ws_bytes,
                                void* stream) {
  if (n < 0 || n_nodes <= 0 || !out_count) return TG_EINVAL;
  if (n > 0 && (!nids || !ts || !out_unique || !out_index)) return TG_EINVAL;
  if (ws_bytes < tg_select_latest_workspace_bytes(n, n_nodes) || !ws) return TG_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int64_t W = (n_nodes + 63) / 64;
  char* p = (char*)ws;
  // flags | best | best_idx first: everything that starts from zero, cleared by ONE memset
  uint8_t* flags = (uint8_t*)p;
  p += align16((size_t)W * 64);
  unsigned long long* best = (unsigned long long*)p;
  p += align16((size_t)n * 8);
  unsigned int* best_idx = (unsigned int*)p;
  p += align16((size_t)n * 4);
  const size_t zero_bytes = (size_t)(p - (char*)ws);
  uint64_t* bm = (uint64_t*)p;
  p += align16((size_t)W * 8);
  uint32_t* rank = (uint32_t*)p;
  p += align16((size_t)(W + 1) * 4);
  hipError_t e = hipMemsetAsync(flags, 0, zero_bytes, st);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_select_latest memset");
    return TG_EHIP;
  }
  if (n > 0) hipLaunchKernelGGL(k_mark_flags, dim3(flat_grid(n, 256)), dim3(256), 0, st, n, nids, flags, n_nodes);
  int rc = unique_compact_launch(flags, bm, n_nodes, rank, out_unique, out_count, n, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, p, scan_ws_bytes(W), st);
  if (rc != TG_OK || n == 0) return rc;
  const unsigned g = flat_grid(n, 256);
  if (ts_is_f64) {
    hipLaunchKernelGGL(k_sel_max<double>, dim3(g), dim3(256), 0, st, n, nids, (const double*)ts, bm, rank, best);
    hipLaunchKernelGGL(k_sel_min<double>, dim3(g), dim3(256), 0, st, n, nids, (const double*)ts, bm, rank, best, best_idx);
  } else {
    hipLaunchKernelGGL(k_sel_max<float>, dim3(g), dim3(256), 0, st, n, nids, (const float*)ts, bm, rank, best);
    hipLaunchKernelGGL(k_sel_min<float>, dim3(g), dim3(256), 0, st, n, nids, (const float*)ts, bm, rank, best, best_idx);
  }
  hipLaunchKernelGGL(k_sel_out, dim3(g), dim3(256), 0, st, out_count, n, best_idx, out_index);
  return check_launch("tg_select_latest");
}
