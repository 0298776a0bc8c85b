// Shared device helpers for libtiger_hip (gfx950 only: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tiger_hip.h"

#define TG_WAVE 64

namespace tg {

void set_hip_error(hipError_t e, const char* what);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_hip_error(e, what);
    return TG_EHIP;
  }
  return TG_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a flat elementwise pass: capped so that long inputs grid-stride
inline unsigned flat_grid(int64_t work_items, int block) {
  int64_t g = cdiv(work_items, block);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (unsigned)g;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (TG_WAVE - 1); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, TG_WAVE);
  return v;
}

// local index of node `id` inside the sorted-unique list encoded by (bitmap, rank)
__device__ __forceinline__ uint32_t bm_rank(const uint64_t* __restrict__ bm, const uint32_t* __restrict__ rank,
                                            int64_t id) {
  const int64_t w = id >> 6;
  const uint64_t below = (1ull << (id & 63)) - 1ull;
  return rank[w] + (uint32_t)__popcll(bm[w] & below);
}

__device__ __forceinline__ bool bm_test(const uint64_t* __restrict__ bm, int64_t id) {
  return (bm[id >> 6] >> (id & 63)) & 1ull;
}

// TimeEncode (time_encoding.py:24-26): the product is rounded to float32 before the
// phase is added (no FMA contraction; the library is also built with -ffp-contract=off),
// and cosf is the accurate OCML routine with full range reduction (never __cosf).
__device__ __forceinline__ float time_enc(float dt, float w, float phi) {
  return cosf(__fadd_rn(__fmul_rn(dt, w), phi));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// order-preserving maps float -> unsigned (for atomicMax on timestamps)
__device__ __forceinline__ uint64_t orderable(double x) {
  uint64_t u = (uint64_t)__double_as_longlong(x);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ uint64_t orderable(float x) {
  uint32_t u = __float_as_uint(x);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return (uint64_t)u;
}


// ---- internal launchers shared between translation units (not part of the C ABI) ----
// sampler with the batch -> query expansion fused in (data_loader.py:79-81,92,128)
int sample_batch_launch(const tg_tcsr* g, int64_t B, const int64_t* src, const int64_t* dst, const int64_t* neg,
                        const double* ts, const int64_t* eids, const int64_t* off, int32_t K, int64_t* nids3,
                        float* ts3f, int64_t* eids_b, int64_t* o_nbr, int64_t* o_eid, float* o_ts, uint8_t* mark,
                        hipStream_t st);
// reprs <- right memory rows of the involved nodes, plus the message/memory time invariants
int consume_gather_check_launch(const tg_model* m, const int64_t* involved, const int32_t* n_involved, int64_t cap,
                                float* reprs, const int64_t* outdated, const int32_t* n_outdated, uint32_t* err,
                                hipStream_t st);
// STEP 4-6 in two launches (phase 0 then 1); the tail work (counts copy, stream offset advance)
// rides on phase 1
struct WritebackArgs {
  int64_t B;
  const int64_t *src, *dst, *eids, *upos, *index;
  const float* ts;          // [>= 2B] float32 event times tiled over cat[src,dst]
  const int32_t* n_upos;
  const float* reprs;       // h(t'+) rows, indexed by local rank (bitmap, rank)
  const uint64_t* bm;
  const uint32_t* rank;
  const float* h;           // [>= 2B, d] embeddings of cat[src,dst]
  uint32_t* err;
  const int32_t* counts_src;  // nullable: 4 ints copied to counts_dst
  int32_t* counts_dst;
  int64_t* offset_dev;        // nullable: += B
};
int writeback_launch(const tg_model* m, const WritebackArgs& a, int phase, hipStream_t st);

}  // namespace tg
