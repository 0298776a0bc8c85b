// T-CSR build on device (reference tiger/data/graph.py:11-42,226-241).
//
// The reference appends, event by event, (dst, eid, t, 0) to the list of src and (src, eid, t, 1) to
// the list of dst, then sorts every list stably by time.  For a time-ordered stream (every JODIE
// file, every stream this package generates; checked by the caller) the stable sort is the identity,
// so the T-CSR order is: entries p = 2e (seen from src) and p = 2e + 1 (seen from dst), sorted
// STABLY by owner node.  That is one least-significant-digit radix sort of the 2E pairs
// (owner, p) on the owner id - 4-bit digits, so the per-thread digit counters of a tile fit LDS
// ([16 digits][256 threads]) and the scatter of a pass is stable by construction (every thread owns 8
// consecutive entries and walks them in order).  Unsorted streams take tg_tcsr_build_host.
#include <algorithm>

#include "tg_step.h"

namespace tg {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;

__device__ __forceinline__ uint32_t owner_of(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, uint32_t p) {
  return (uint32_t)((p & 1u) ? dst[p >> 1] : src[p >> 1]);
}

// per-tile digit counts with their in-tile exclusive prefix over threads; returns nothing, leaves
// cnt[d][t] = number of entries with digit d owned by threads < t of this tile
__device__ __forceinline__ void tile_prefix(uint32_t (&cnt)[16][RS_THREADS + 1], const uint32_t* keys, int n_items,
                                            int shift, uint32_t* totals) {
  const int t = threadIdx.x;
#pragma unroll
  for (int d = 0; d < 16; ++d) cnt[d][t] = 0;
  for (int i = 0; i < n_items; ++i) cnt[(keys[i] >> shift) & 15u][t] += 1;
  __syncthreads();
  // 16 digits x 256 threads: wave w scans digits 4w .. 4w+3, lane l owns threads 4l .. 4l+3
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const int d = wave * 4 + dd;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = cnt[d][lane * 4 + j];
      s += v[j];
    }
    uint32_t incl = s;  // inclusive scan of s across the 64 lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    uint32_t run = incl - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cnt[d][lane * 4 + j] = run;
      run += v[j];
    }
    if (lane == 63) totals[d] = incl;
  }
  __syncthreads();
}

// pass kernels.  FIRST: the key is computed from (src, dst), the payload is p itself.
template <bool FIRST>
__global__ void __launch_bounds__(RS_THREADS) k_rs_count(uint32_t P, const uint32_t* __restrict__ keys_in,
                                                         const int64_t* __restrict__ src,
                                                         const int64_t* __restrict__ dst, int shift,
                                                         uint32_t* __restrict__ hist, uint32_t nblocks) {
  __shared__ uint32_t cnt[16][RS_THREADS + 1];
  __shared__ uint32_t totals[16];
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE + threadIdx.x * RS_ITEMS;  // 64-bit: P may be close to 2^32
  uint32_t keys[RS_ITEMS];
  int n = 0;
  for (int i = 0; i < RS_ITEMS; ++i) {
    const uint64_t p = base + i;
    if (p < P) keys[n++] = FIRST ? owner_of(src, dst, (uint32_t)p) : keys_in[p];
  }
  tile_prefix(cnt, keys, n, shift, totals);
  if (threadIdx.x < 16) hist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = totals[threadIdx.x];
}

template <bool FIRST>
__global__ void __launch_bounds__(RS_THREADS) k_rs_scatter(uint32_t P, const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in,
                                                           const int64_t* __restrict__ src,
                                                           const int64_t* __restrict__ dst, int shift,
                                                           const uint32_t* __restrict__ hist, uint32_t nblocks,
                                                           uint32_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t cnt[16][RS_THREADS + 1];
  __shared__ uint32_t totals[16];
  __shared__ uint32_t gbase[16];
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE + threadIdx.x * RS_ITEMS;
  uint32_t keys[RS_ITEMS], vals[RS_ITEMS];
  int n = 0;
  for (int i = 0; i < RS_ITEMS; ++i) {
    const uint64_t p = base + i;
    if (p < P) {
      keys[n] = FIRST ? owner_of(src, dst, (uint32_t)p) : keys_in[p];
      vals[n] = FIRST ? (uint32_t)p : vals_in[p];
      ++n;
    }
  }
  if (threadIdx.x < 16) gbase[threadIdx.x] = hist[(uint64_t)threadIdx.x * nblocks + blockIdx.x];  // scanned: global start
  tile_prefix(cnt, keys, n, shift, totals);
  const int t = threadIdx.x;
  for (int i = 0; i < n; ++i) {  // in entry order: the pass is stable
    const uint32_t d = (keys[i] >> shift) & 15u;
    const uint32_t pos = gbase[d] + cnt[d][t]++;
    keys_out[pos] = keys[i];
    vals_out[pos] = vals[i];
  }
}

// exclusive scan, three launches per level: block scans + block sums, scan of the sums (recursive), add
template <typename T>
__global__ void __launch_bounds__(256) k_scan_tiles(T* __restrict__ a, int64_t n, T* __restrict__ sums) {
  __shared__ T sh[256];
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  T v[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = base + j < n ? a[base + j] : (T)0;
    s += v[j];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const T up = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : (T)0;
    __syncthreads();
    sh[threadIdx.x] += up;
    __syncthreads();
  }
  T run = sh[threadIdx.x] - s;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < n) a[base + j] = run;
    run += v[j];
  }
  if (threadIdx.x == 255 && sums) sums[blockIdx.x] = sh[255];
}
template <typename T>
__global__ void k_scan_add(T* __restrict__ a, int64_t n, const T* __restrict__ sums) {
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const T off = sums[blockIdx.x];
  for (int j = 0; j < 4; ++j) {
    const int64_t k = i + j * 256;
    if (k < n) a[k] += off;
  }
}
template <typename T>
static void exclusive_scan(T* a, int64_t n, T* scratch, hipStream_t st) {
  const int64_t nb = cdiv(n, 1024);
  hipLaunchKernelGGL(k_scan_tiles<T>, dim3((unsigned)nb), dim3(256), 0, st, a, n, nb > 1 ? scratch : (T*)nullptr);
  if (nb > 1) {
    exclusive_scan(scratch, nb, scratch + nb, st);
    hipLaunchKernelGGL(k_scan_add<T>, dim3((unsigned)nb), dim3(256), 0, st, a, n, scratch);
  }
}
static size_t scan_scratch_elems(int64_t n) {
  size_t total = 0;
  for (int64_t nb = cdiv(n, 1024); nb > 1; nb = cdiv(nb, 1024)) total += (size_t)nb;
  return total + 8;
}

__global__ void k_degree(int64_t E, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                         unsigned long long* __restrict__ deg) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    atomicAdd(deg + src[i], 1ull);
    atomicAdd(deg + dst[i], 1ull);
  }
}

__global__ void k_tcsr_fill(uint32_t P, const uint32_t* __restrict__ order, const int64_t* __restrict__ src,
                            const int64_t* __restrict__ dst, const double* __restrict__ ts,
                            const int64_t* __restrict__ eid, double* __restrict__ ts_out, int32_t* __restrict__ nbr_out,
                            int32_t* __restrict__ eid_out) {
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < P; s += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t p = order[s], e = p >> 1;
    ts_out[s] = ts[e];
    nbr_out[s] = (int32_t)((p & 1u) ? src[e] : dst[e]);
    eid_out[s] = (int32_t)((uint32_t)eid[e] | ((p & 1u) << 31));
  }
}

}  // namespace tg

using namespace tg;

static int key_bits(int64_t num_node) {
  int b = 1;
  while (((int64_t)1 << b) < num_node) ++b;
  return (b + 3) / 4 * 4;
}

extern "C" size_t tg_tcsr_build_device_workspace_bytes(int64_t num_events, int64_t num_node) {
  if (num_events < 0 || num_node <= 0) return 0;
  const size_t P = 2 * (size_t)num_events, nblocks = (P + RS_TILE - 1) / RS_TILE;
  return align16(P * 4) * 4 + align16(16 * nblocks * 4) + align16(scan_scratch_elems(16 * nblocks) * 4) +
         align16(scan_scratch_elems(num_node + 1) * 8) + 256;
}

extern "C" int tg_tcsr_build_device(int64_t E, const int64_t* src, const int64_t* dst, const double* ts,
                                    const int64_t* eid, int64_t num_node, int64_t* indptr, double* ts_out,
                                    int32_t* nbr_out, int32_t* eid_out, void* ws, size_t ws_bytes, void* stream) {
  if (E < 0 || num_node <= 0 || num_node > 0x7fffffffLL || 2 * E > 0xffffffffLL) return TG_EINVAL;
  if (!indptr) return TG_EINVAL;
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(indptr, 0, (size_t)(num_node + 1) * sizeof(int64_t), st);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_tcsr_build_device memset");
    return TG_EHIP;
  }
  if (E == 0) return TG_OK;
  if (!src || !dst || !ts || !eid || !ts_out || !nbr_out || !eid_out) return TG_EINVAL;
  const uint32_t P = (uint32_t)(2 * E);
  const uint32_t nblocks = (P + RS_TILE - 1) / RS_TILE;
  Carver cv(ws, ws_bytes);
  uint32_t* k0 = cv.take<uint32_t>(P);
  uint32_t* v0 = cv.take<uint32_t>(P);
  uint32_t* k1 = cv.take<uint32_t>(P);
  uint32_t* v1 = cv.take<uint32_t>(P);
  uint32_t* hist = cv.take<uint32_t>((size_t)16 * nblocks);
  uint32_t* hscr = cv.take<uint32_t>(scan_scratch_elems((int64_t)16 * nblocks));
  int64_t* iscr = cv.take<int64_t>(scan_scratch_elems(num_node + 1));
  if (!cv.ok) return TG_EWORKSPACE;
  // indptr: degree histogram shifted by one, then exclusive scan of [deg(0), deg(1), ...] in place
  hipLaunchKernelGGL(k_degree, dim3(flat_grid(E, 256)), dim3(256), 0, st, E, src, dst, (unsigned long long*)indptr);
  exclusive_scan<int64_t>(indptr, num_node + 1, iscr, st);
  // stable LSD radix sort of (owner, p) on the owner id
  const int bits = key_bits(num_node);
  uint32_t *ki = k0, *vi = v0, *ko = k1, *vo = v1;
  for (int shift = 0; shift < bits; shift += 4) {
    const bool first = shift == 0;
    if (first)
      hipLaunchKernelGGL(k_rs_count<true>, dim3(nblocks), dim3(RS_THREADS), 0, st, P, ki, src, dst, shift, hist, nblocks);
    else
      hipLaunchKernelGGL(k_rs_count<false>, dim3(nblocks), dim3(RS_THREADS), 0, st, P, ki, src, dst, shift, hist, nblocks);
    exclusive_scan<uint32_t>(hist, (int64_t)16 * nblocks, hscr, st);
    if (first)
      hipLaunchKernelGGL(k_rs_scatter<true>, dim3(nblocks), dim3(RS_THREADS), 0, st, P, ki, vi, src, dst, shift, hist,
                         nblocks, ko, vo);
    else
      hipLaunchKernelGGL(k_rs_scatter<false>, dim3(nblocks), dim3(RS_THREADS), 0, st, P, ki, vi, src, dst, shift, hist,
                         nblocks, ko, vo);
    std::swap(ki, ko);
    std::swap(vi, vo);
  }
  hipLaunchKernelGGL(k_tcsr_fill, dim3(flat_grid(P, 256)), dim3(256), 0, st, P, vi, src, dst, ts, eid, ts_out, nbr_out,
                     eid_out);
  return check_launch("tg_tcsr_build_device");
}
