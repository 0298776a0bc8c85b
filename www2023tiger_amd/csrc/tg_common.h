// Shared device helpers for libtiger_hip (gfx950 only: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

// ---- kernel-bound timing (tg_profiler, eager launches only) ----------------------------------------------------------
// A HIP-event pair AROUND a launch measures the launch plus the two event records (~4.5 us on this stack); an event pair
// BOUND to the dispatch (hipExtLaunchKernelGGL's start / stop events: the dispatch's own begin / end timestamps) measures
// the kernel as rocprofv3 does.  While tg_stream_step runs with a profiler attached, the launches of the step's main
// kernels go through TG_KLAUNCH with a slot selected by the caller (KSlot): the last launch made under a slot is timed
// and its kernel's name (the launch expression) recorded.  Without a profiler TG_KLAUNCH is hipLaunchKernelGGL.
enum KtSlot : int { KT_NONE = -1, KT_COLLATE = 0, KT_CORE, KT_FC1, KT_FC2, KT_UPDATER, KT_QROWS, KT_WRITEBACK, KT_GATHER, KT_COUNT };
struct KTimer {
  hipEvent_t ev[KT_COUNT][2];
  const char* name[KT_COUNT];
  bool hit[KT_COUNT];
};
extern thread_local KTimer* g_kt;
extern thread_local int g_kt_slot;
struct KSlot {  // selects the slot for the launches made in its scope
  int prev;
  explicit KSlot(int s) : prev(g_kt_slot) { g_kt_slot = s; }
  ~KSlot() { g_kt_slot = prev; }
};
#define TG_KLAUNCH(kern, grid, block, shmem, st, ...)                                                               \
  do {                                                                                                              \
    ::tg::KTimer* kt__ = ::tg::g_kt;                                                                                \
    const int ks__ = ::tg::g_kt_slot;                                                                               \
    if (kt__ && ks__ >= 0) {                                                                                        \
      hipExtLaunchKernelGGL(kern, grid, block, shmem, st, kt__->ev[ks__][0], kt__->ev[ks__][1], 0, __VA_ARGS__);    \
      kt__->name[ks__] = #kern;                                                                                     \
      kt__->hit[ks__] = true;                                                                                       \
    } else {                                                                                                        \
      hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                                                \
    }                                                                                                               \
  } while (0)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a flat elementwise pass: capped so that long inputs grid-stride
inline unsigned flat_grid(int64_t work_items, int block) {
  int64_t g = cdiv(work_items, block);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (unsigned)g;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (TG_WAVE - 1); }

// Sixteen bytes of zeros in device memory (zero_line(): its address, a plain global pointer handed to kernels as an
// argument).  Where lanes past the end of a row must read zeros they LOAD them from there - the load stays unconditional
// and its result needs no select.  (A load inside a divergent branch, or a select on a loaded value that the scheduler
// places right behind the load, makes the compiler wait for every load in flight: a ring of gathers then hides nothing.
// Taking the symbol's address in device code instead turns the selected pointer into a flat one: flat loads, which count
// on both counters and are waited for one by one.)
const float* zero_line();

// Wave-wide sum, result broadcast to every lane.  DPP row shifts / row broadcasts (GFX9
// encodings, valid on gfx950) instead of six ds_bpermute round trips through the LDS unit.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row total
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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

// row of node v in this process's state tables (tiger_hip.h: tg_model.row_of; identity unless the state is physically
// partitioned)
__device__ __forceinline__ int64_t state_row(const tg_model& m, int64_t v) { return m.row_of ? (int64_t)m.row_of[v] : v; }

// TimeEncode (time_encoding.py:24-26): the product is rounded to float32 before the
// phase is added (no FMA contraction; the library is also built with -ffp-contract=off),
// and the cosine is accurate over the whole argument range (never __cosf).
// cos(x) for |x| <= 3e6: quadrant n = rint(x * 2/pi) (n < 2^21), three-term Cody-Waite
// reduction with explicit fma (exact products), Cephes minimax polynomials on [-pi/4, pi/4].
// Max abs error 9.4e-8 against float64 cos over [-3e6, 3e6] (about 3x fewer instructions
// than OCML's cosf, whose large-argument path dominated the attention gather kernel).
__device__ __forceinline__ float cos_cw(float x) {
  const float n = rintf(__fmul_rn(x, 0.6366197723675814f));
  float r = fmaf(-n, 1.5707963705062866f, x);
  r = fmaf(-n, -4.371138828673793e-08f, r);
  r = fmaf(-n, -1.7763568394002505e-15f, r);
  const float z = __fmul_rn(r, r);
  const float s = fmaf(__fmul_rn(r, z), fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float c = fmaf(__fmul_rn(z, z), fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                       fmaf(z, -0.5f, 1.0f));
  const int q = (int)n & 3;
  const float v = (q & 1) ? s : c;
  return (q == 1 || q == 2) ? -v : v;
}

// Large arguments (|x| > 3e6, i.e. time gaps beyond ~35 days at the highest frequency): the same
// quadrant scheme with the reduction carried out in float64 (two-term pi/2, exact products via
// fma; n < 2^53 is exact up to |x| ~ 1e16, far beyond any timestamp).  This replaces OCML's
// cosf/sinf, whose Payne-Hanek path - even when never taken - costs every kernel that inlines it
// ~60 VGPRs and scratch: k_attn_core ran 44 us with it and 24 us without.
__device__ __forceinline__ void reduce_pio2_f64(float x, float& r, int& q) {
  const double xd = (double)x;
  const double n = rint(xd * 0.63661977236758134308);
  double rd = fma(-n, 1.57079632679489655800e+00, xd);
  rd = fma(-n, 6.12323399573676603587e-17, rd);
  r = (float)rd;
  q = (int)((long long)n & 3);
}
__device__ __forceinline__ void sincos_poly(float r, float& s, float& c) {
  const float z = __fmul_rn(r, r);
  s = fmaf(__fmul_rn(r, z), fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  c = fmaf(__fmul_rn(z, z), fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
           fmaf(z, -0.5f, 1.0f));
}
__device__ __forceinline__ float cos_big(float x) {
  float r, s, c;
  int q;
  reduce_pio2_f64(x, r, q);
  sincos_poly(r, s, c);
  const float v = (q & 1) ? s : c;
  return (q == 1 || q == 2) ? -v : v;
}
__device__ __forceinline__ float sin_big(float x) {
  float r, s, c;
  int q;
  reduce_pio2_f64(x, r, q);
  sincos_poly(r, s, c);
  const float v = (q & 1) ? c : s;
  return (q >= 2) ? -v : v;
}

__device__ __forceinline__ float time_enc(float dt, float w, float phi) {
  const float x = __fadd_rn(__fmul_rn(dt, w), phi);
  return fabsf(x) <= 3.0e6f ? cos_cw(x) : cos_big(x);
}

// The attention core evaluates K*d of these per centre and is VALU-bound on them: there the cosine of the SAME float32
// argument comes from the hardware unit (v_cos_f32, input in revolutions) behind a three-term Cody-Waite reduction
// modulo 2*pi - 3.5e-7 max abs error against float64 cos on |x| <= 3e6 (tools/micro/vcos_err.hip; the polynomial
// path: 9.2e-8), about ten issue slots instead of twenty-five.  k_attn_core is the forward of every path (streaming,
// the operator path, evaluation and the training step), so all of them encode the KEYS with the hardware cosine; what is
// compared element-wise with the reference's encoding (tg_time_encode, the raw messages of STEP 5) and the backward
// pass's sin / cos keep the polynomial path.  The 3.5e-7 sit two orders below the 1e-4 parity bar.
__device__ __forceinline__ float cos_hw(float x) {
  const float n = rintf(__fmul_rn(x, 0.15915494309189535f));
  float r = fmaf(-n, 6.2831854820251465f, x);
  r = fmaf(-n, -1.7484555314695172e-07f, r);
  r = fmaf(-n, -7.1054273576010019e-15f, r);
  return __builtin_amdgcn_cosf(__fmul_rn(r, 0.15915494309189535f));
}
__device__ __forceinline__ float time_enc_fast(float dt, float w, float phi) {
  const float x = __fadd_rn(__fmul_rn(dt, w), phi);
  return fabsf(x) <= 3.0e6f ? cos_hw(x) : cos_big(x);
}

// sin with the same three-term Cody-Waite reduction as cos_cw above
__device__ __forceinline__ float sin_cw(float x) {
  const float n = rintf(__fmul_rn(x, 0.6366197723675814f));
  float r = fmaf(-n, 1.5707963705062866f, x);
  r = fmaf(-n, -4.371138828673793e-08f, r);
  r = fmaf(-n, -1.7763568394002505e-15f, r);
  const float z = __fmul_rn(r, r);
  const float s = fmaf(__fmul_rn(r, z), fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float c = fmaf(__fmul_rn(z, z), fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                       fmaf(z, -0.5f, 1.0f));
  const int q = (int)n & 3;
  const float v = (q & 1) ? c : s;
  return (q >= 2) ? -v : v;
}
__device__ __forceinline__ float time_enc_sin(float dt, float w, float phi) {
  const float x = __fadd_rn(__fmul_rn(dt, w), phi);
  return fabsf(x) <= 3.0e6f ? sin_cw(x) : sin_big(x);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// gate nonlinearities of the GRU epilogue on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp);
// arguments are bounded pre-activations, so the fast forms stay ~1e-6 relative - two orders
// below the 1e-4 parity budget - at a third of the instructions of expf / tanhf
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  const float t = (1.0f - e) * __frcp_rn(1.0f + e);
  return copysignf(t, x);
}

// ---- dropout (training only).  Counter-based: element `idx` of mask stream `stream` at step
// `rng[1]` of seed `rng[0]` is kept iff hash >= p * 2^32; kept values are scaled by 1/(1-p).  No
// state is carried between kernels, so forward and backward regenerate identical masks, and the
// host can reproduce them (tests/_util.py::dropout_keep).
struct DropCfg {
  float p;              // 0: dropout off
  float scale;          // 1 / (1 - p)
  uint32_t thresh;      // p * 2^32
  const uint64_t* rng;  // device: {seed, step counter}
};
enum { DROP_ATTN = 1, DROP_SCORE = 2, DROP_SEQ_ATTN = 3, DROP_SEQ_MERGER = 4 };
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint64_t drop_key(const DropCfg& c) {
  return c.rng ? c.rng[0] + c.rng[1] * 0x9E3779B97F4A7C15ull : 0ull;
}
__device__ __forceinline__ bool drop_keep(uint64_t key, uint32_t stream, uint64_t idx, uint32_t thresh) {
  uint32_t h = mix32((uint32_t)idx ^ (uint32_t)key);
  h = mix32(h + (uint32_t)(idx >> 32) * 0x9e3779b9u + (uint32_t)(key >> 32) + stream * 0x85ebca6bu);
  return h >= thresh;
}
inline DropCfg make_drop(float p, const uint64_t* rng) {
  DropCfg c{};
  if (p > 0.f && rng) {
    c.p = p;
    c.scale = 1.0f / (1.0f - p);
    const double t = (double)p * 4294967296.0;
    c.thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    c.rng = rng;
  }
  return c;
}

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
struct CentresRider;
int sample_batch_launch(const tg_tcsr* g, int64_t B, const int64_t* src, const int64_t* dst, const int64_t* neg,
                        const double* ts, const int64_t* eids, const int64_t* off, int32_t K, int64_t* nids3,
                        float* ts3f, int64_t* eids_b, int64_t* o_nbr, int64_t* o_eid, float* o_ts, uint8_t* mark,
                        hipStream_t st, uint32_t* tmin_key = nullptr, const CentresRider* rider = nullptr);
// recent-edges sampler over float32 query times (the second hop of --n_layers 2 is sampled at the neighbours' own
// float32 timestamps, data_loader.py:131); marks the sampled ids in `mark` when given
// tg_restart_apply on the first *n_dev (nullable: all n) entries of the list (tg_memory.hip)
int restart_apply_dev(const tg_model* m, int64_t n, const int64_t* nids, const float* h_left, const float* h_right,
                      const float* prev_ts, const int32_t* n_dev, hipStream_t st);
int sample_edges_f32_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const float* ts, int32_t K, int64_t* o_nbr,
                            int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st);
int sample_nodes_f32_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const float* ts, int32_t K, int64_t* o_nbr,
                            int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st);
// recent-nodes sampler (graph.py:129-143) over prepared query arrays; marks queries and neighbours in `mark` when given
int sample_nodes_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const double* ts, int32_t K, int64_t* o_nbr,
                        int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st);
// uniform sampler (graph.py:101-115) over prepared query arrays, the graph's MT19937 state on the device
int sample_uniform_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const double* ts, int32_t K, uint32_t* mt_state,
                          int64_t* o_nbr, int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st);
int sample_uniform_f32_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const float* ts, int32_t K, uint32_t* mt_state,
                              int64_t* o_nbr, int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st);
// the lazy-restart loop body of train_self_supervised.py:152-163 with the static restarter (tiger_hip.h: tg_lazy_restart);
// runs between the sampler (flags, *tmin_key) and the compaction
// rlist / rlist32 (nullable, static form only): also list the re-initialised nodes and rewrite their rows of m->c_table
int lazy_restart_launch(const tg_tcsr* g, const tg_model* m, const tg_lazy_restart* lz, const uint8_t* flags,
                        const uint32_t* tmin_key, int32_t* n_restarted, hipStream_t st, int64_t* rlist = nullptr,
                        int32_t* rlist32 = nullptr);
// positive-node dedup of the fused step (select_latest_nids on float32 ts): best[rank(node)] =
// max over positions of (ts_key << 32 | ~pos), then the winners.  The two passes ride on other
// launches of the step (they are ~2B threads of work each, not worth a launch of their own).
struct PosArgs {
  int64_t B;
  const int64_t* nids3;  // cat[src, dst, ...]
  const float* ts;       // [>= B]
  const uint64_t* bm;
  const uint32_t* rank;
  unsigned long long* best;
  int32_t* count;
  int64_t *upos, *index;
  int32_t* upos32;  // nullable: the winners' node ids once more as int32 (output rows of the eager updater launch)
  uint32_t* chk_err;  // lean steps: the core launch checks the time invariants of every neighbour with a pending message
  int64_t* advance_off;  // lean embed-only steps: the core launch advances the stream offset by B when it is done with it
  // nullable: [2B] the winners pass also leaves, for EVERY position of cat[src, dst], the node whose STEP 6 row it is
  // (the winner of its node) or -1 - the scatter list of the product that writes h(t-) straight into the left memory
  // (write-back rider, below)
  int32_t* win_row;
  // nullable (split updater, tg_step.h: GruSplit): per winner slot the position of the event's OTHER endpoint in
  // cat[src, dst] and the event's edge id - with `index` the row indices of the four segments of the winner's raw message
  // [snap[index] | snap[oth] | efeat[weid] | snap_te[index]] (memory.py:89-106), gathered by the product that forms W_ih msg
  const int64_t* eids;  // [B] edge ids of the batch
  int64_t *oth, *weid;
};
// dedup slot of a node: its rank in the involved set, or (lean steps: no involved set) the node id itself
__device__ __forceinline__ int64_t pos_slot(const PosArgs& a, int64_t node) {
  return a.bm ? (int64_t)bm_rank(a.bm, a.rank, node) : node;
}
// where the ids / times of cat[src, dst, neg] come from: the step's own copy (nids3, float32 times), or - for work that
// shares a launch with the sampler, which is still writing that copy - the batch arrays themselves
struct ArrayIds {
  const int64_t* nids3;
  const float* ts3f;
  __device__ __forceinline__ int64_t id(int64_t i) const { return nids3[i]; }
  __device__ __forceinline__ float tf(int64_t e) const { return ts3f[e]; }
};
struct RawIds {
  const int64_t *src, *dst, *neg;
  const double* ts;
  int64_t off, B;
  __device__ __forceinline__ int64_t id(int64_t i) const {
    const int64_t e = i % B;
    const int r = (int)(i / B);
    return r == 0 ? src[off + e] : (r == 1 ? dst[off + e] : neg[off + e]);
  }
  __device__ __forceinline__ float tf(int64_t e) const { return (float)ts[off + e]; }
};
template <class Ids>
__device__ __forceinline__ void pos_max_pass(const PosArgs& a, const Ids& b, int64_t tid, int64_t nth) {
  if (tid == 0) *a.count = 0;  // the winners pass (a later launch) counts into it
  for (int64_t i = tid; i < 2 * a.B; i += nth) {
    const int64_t e = i < a.B ? i : i - a.B;
    const unsigned long long key = (orderable(b.tf(e)) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
    atomicMax(a.best + pos_slot(a, b.id(i)), key);
  }
}
__device__ __forceinline__ void pos_max_pass(const PosArgs& a, int64_t tid, int64_t nth) {
  pos_max_pass(a, ArrayIds{a.nids3, a.ts}, tid, nth);
}
__device__ __forceinline__ void pos_winners_pass(const PosArgs& a, int64_t tid, int64_t nth) {
  for (int64_t i = tid; i < 2 * a.B; i += nth) {
    const int64_t e = i < a.B ? i : i - a.B;
    const int64_t node = a.nids3[i];
    const unsigned long long key = (orderable(a.ts[e]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
    const bool win = a.best[pos_slot(a, node)] == key;
    if (win) {
      const int slot = atomicAdd(a.count, 1);
      a.upos[slot] = node;
      a.index[slot] = i;
      if (a.upos32) a.upos32[slot] = (int32_t)node;
      if (a.oth) {
        a.oth[slot] = i < a.B ? i + a.B : i - a.B;
        a.weid[slot] = a.eids[e];
      }
    }
    if (a.win_row) a.win_row[i] = win ? (int32_t)node : -1;
  }
}
// Eager updates, direct form: the centre row is read from the state tables themselves,
//   c_i = (has_msg[v] ? pending[v] : right[v]) + nfeat[v],   v = nid_i
// (what reprs[local(v)] holds after STEP 1-2, tiger.py:214-221), so no compact copy of the involved rows is made.
// The launch also carries what rode on the gather launch: the time invariants of compute_messages over the outdated
// list (message_modules.py:158-159, tiger.py:325-327) and the first dedup pass.
struct DirectArgs {
  const int64_t* outdated;
  const int32_t* n_outdated;
  int64_t cap;
  uint32_t* err;
  // snapshot for the one-launch write-back (nullable): for position i < n_snap of cat[src, dst] the message-source
  // memory row of its node as STEP 5 will want it (tiger.py:422-442: + node features; msg_src = right: the right memory
  // as STEP 4 leaves it, i.e. the centre row itself) and that memory's time.  Taken here, before anything is written,
  // it lets STEP 5 share a launch with STEP 6, which overwrites those very rows of the left memory.
  float4* snap;
  float* snap_ts;
  int64_t n_snap;
  // lean step (no outdated list): the invariants are checked per centre here and per neighbour in the core launch -
  // the same node set, involved & has-message, some nodes more than once
  int per_row_checks;
  // nullable, with snap: TE(t_i - snap_ts[i]) of the same positions (the time segment of the raw message position i would
  // store in STEP 5, memory.py:89-106; the same rounding as the mailbox row: time_enc) - the split updater multiplies the
  // messages of the winning positions by W_ih while the attention block is still running (tg_step.h: GruSplit)
  float4* snap_te;
};
// (`id`: the node's ROW in the state tables, state_row)
__device__ __forceinline__ void check_msg_times(const tg_model& m, int64_t id, uint32_t* err) {
  const float mts = m.msg_ts[id], last = (m.msg_src == TG_SRC_LEFT ? m.left_ts : m.right_ts)[id];
  if (last > mts) atomicOr(err, TG_ERR_MSG_BEFORE_MEM);
  if (m.msg_src == TG_SRC_LEFT && !(mts == last)) atomicOr(err, TG_ERR_MSG_TS_MISMATCH);
}
// thread tid of nth: Q centre rows (float4 granularity), the checks and the first dedup pass
template <class Ids>
__device__ __forceinline__ void centres_direct_body(const tg_model& m, int64_t Q, const Ids& ids,
                                                    const float4* __restrict__ nf, float4* __restrict__ out,
                                                    const DirectArgs& da, const PosArgs& pos, int64_t tid, int64_t nth) {
  const int d4 = m.d / 4;
  const float4* right = reinterpret_cast<const float4*>(m.right_vals);
  const float4* pend = reinterpret_cast<const float4*>(m.pending_vals);
  // without a copy of the centre rows (out == nullptr) only the snapshot positions need their rows walked; the other
  // centres keep their time-invariant check (one thread per centre, below)
  const int64_t rows_full = out ? Q : min(Q, da.snap ? da.n_snap : (int64_t)0);
  const int64_t total = rows_full * d4;
  if (da.per_row_checks) {
    for (int64_t i = rows_full + tid; i < Q; i += nth) {
      const int64_t r = state_row(m, ids.id(i));
      if (bm_test(m.has_msg, r)) check_msg_times(m, r, da.err);
    }
  }
  // U elements per thread in flight: with fewer threads than elements (the centres as riders of another launch) the
  // three dependent round trips per element (id -> has-message bit -> row) are paid once per U elements, not per element
  constexpr int U = 4;
  for (int64_t t0 = tid; t0 < total; t0 += U * nth) {
    int64_t i[U], id[U], r[U];
    int c[U];
    bool live[U], pending[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = t0 + u * nth;
      live[u] = t < total;
      const int64_t tc = live[u] ? t : tid;
      i[u] = tc / d4;
      c[u] = (int)(tc - i[u] * d4);
      id[u] = ids.id(i[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      r[u] = state_row(m, id[u]);  // state by row, features by node id
      pending[u] = bm_test(m.has_msg, r[u]);
    }
    float4 v[U], f[U], l[U];
    // (out == nullptr - the centre rows live in a per-node table, tg_model.c_table: only the snapshot rows are read)
    const bool left_snap = da.snap && m.msg_src == TG_SRC_LEFT;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool snap_u = da.snap && i[u] < da.n_snap;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (out || (snap_u && !left_snap)) v[u] = (pending[u] ? pend : right)[r[u] * d4 + c[u]];
      f[u] = (nf && (out || snap_u)) ? nf[id[u] * d4 + c[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
      l[u] = v[u];
      if (snap_u && left_snap) l[u] = reinterpret_cast<const float4*>(m.left_vals)[r[u] * d4 + c[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
      const int64_t t = t0 + u * nth;
      v[u].x += f[u].x; v[u].y += f[u].y; v[u].z += f[u].z; v[u].w += f[u].w;
      if (out) out[t] = v[u];
      if (da.per_row_checks && c[u] == 0 && pending[u]) check_msg_times(m, r[u], da.err);
      if (da.snap && i[u] < da.n_snap) {
        float mem_ts;
        if (m.msg_src == TG_SRC_LEFT) {
          l[u].x += f[u].x; l[u].y += f[u].y; l[u].z += f[u].z; l[u].w += f[u].w;
          da.snap[t] = l[u];
          mem_ts = m.left_ts[r[u]];
        } else {
          da.snap[t] = v[u];
          mem_ts = pending[u] ? m.msg_ts[r[u]] : m.right_ts[r[u]];
        }
        if (c[u] == 0) da.snap_ts[i[u]] = mem_ts;
        if (da.snap_te) {
          const int64_t hb = da.n_snap >> 1, ev = i[u] < hb ? i[u] : i[u] - hb;  // (n_snap = 2B positions of cat[src, dst])
          const float dt = ids.tf(ev) - mem_ts;
          const float4 w = reinterpret_cast<const float4*>(m.te_freq)[c[u]], q = reinterpret_cast<const float4*>(m.te_phase)[c[u]];
          da.snap_te[t] = make_float4(time_enc(dt, w.x, q.x), time_enc(dt, w.y, q.y), time_enc(dt, w.z, q.z), time_enc(dt, w.w, q.w));
        }
      }
    }
  }
  if (da.outdated) {
    const int64_t no = min((int64_t)*da.n_outdated, da.cap);
    for (int64_t i = tid; i < no; i += nth) check_msg_times(m, da.outdated[i], da.err);
  }
  if (pos.best) pos_max_pass(pos, ids, tid, nth);
}
// The centres of a lean step depend on nothing the sampler produces (ids and times come from the batch arrays, the dedup
// slots are node ids), so they ride on the sampler's launch as `blocks` extra workgroups at the end of its grid.
struct CentresRider {
  tg_model m;
  const float4* nf;
  float4* out;
  DirectArgs da;
  PosArgs pos;
  unsigned blocks;
};
// reprs <- right memory rows of the involved nodes, plus the message/memory time invariants
// (+ the first dedup pass when pos != nullptr)
// eager: rows of nodes with a pending message come from m->pending_vals (see tiger_hip.h) instead of being left
// for the updater launch to fill
int consume_gather_check_launch(const tg_model* m, const int64_t* involved, const int32_t* n_involved, int64_t cap,
                                float* reprs, const int64_t* outdated, const int32_t* n_outdated, uint32_t* err,
                                hipStream_t st, const PosArgs* pos = nullptr, bool eager = false);
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
  // multi-GPU write-back: rows come from the all-gathered buffer instead of (reprs, h):
  // position idx of cat[src,dst] reads row new_row[po + idx] / left_row[po + idx] of `rows`,
  // po = 2 * *plan_off (the plan arrays are resident for the whole stream)
  const float* rows;
  const int64_t *new_row, *left_row;
  const int64_t* plan_off;
  // self-cleaning of the step workspace by the last kernel (nullable): flags[0, flag_bytes) = 0,
  // best[0, counts[0]) = 0, counts[0..3] = 0 (after the copy to counts_dst)
  uint8_t* clean_flags;
  int64_t flag_bytes;
  unsigned long long* clean_best;
  int32_t* clean_counts;
  int64_t* lazy_batch;  // nullable: the lazy restart's batch counter, += 1 by the last kernel of the step
  // partitioned state: only nodes with owner[node] == my_rank are written / checked; STEP 4 may read the
  // owner's table of precomputed updater rows
  const int32_t* owner;
  int32_t my_rank;
  int32_t new_from_pending;
  // one-launch write-back (phase 2): STEP 5 reads these pre-batch copies instead of the tables STEP 6 overwrites:
  // snap[i] = message-source memory row (+ node features) of position i of cat[src, dst], snap_ts[i] its time
  const float* snap;
  const float* snap_ts;
  // lean step (no involved set): the dedup slots are indexed by node id - the slots of the batch's positive nodes are
  // zeroed, not a prefix - and counts[0], counts[1] are reported as -1
  int clean_best_by_pos;
};
// phase 0 / 1: the two launches of tg_memory.hip's hazard analysis; phase 2: STEP 4-6 in one launch (needs a.snap)
int writeback_launch(const tg_model* m, const WritebackArgs& a, int phase, hipStream_t st);

// system-scope 16-byte accesses (tg_part.h: windows other GPUs store into)
__device__ __forceinline__ void st_sys(float4* p, float4 v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  const unsigned long long a = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
  const unsigned long long b = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
  __hip_atomic_store(q, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(q + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float4 ld_sys(const float4* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                     __uint_as_float((unsigned)(b >> 32)));
}

// ---- fused write-back (tiger.py:229-255).  Hazards: STEP 5 reads the message memory
// rows of BOTH endpoints, STEP 4 writes right-memory rows and STEP 6 left-memory rows of
// other nodes in other waves, so a kernel boundary must separate STEP 5 from whichever
// step writes the message memory:  msg_src=left  -> [4 + 5] | [6],  msg_src=right -> [4] | [5 + 6].
// (`id`, `own`, `other` below: ROWS of the state tables - state_row(node) - where the tables are physically partitioned;
// feature tables are addressed by node id)
__device__ __forceinline__ void wb_step4(const tg_model& m, int64_t id, int64_t u, const float4* __restrict__ reprs,
                                         uint32_t* err, int lane) {
  if (!bm_test(m.has_msg, id)) return;  // wave-uniform
  const int w4 = m.d / 4;
  float4* right = reinterpret_cast<float4*>(m.right_vals);
  for (int c = lane; c < w4; c += TG_WAVE) right[id * w4 + c] = reprs[u * w4 + c];
  if (lane == 0) {
    const float mts = m.msg_ts[id];
    if (m.right_ts[id] > mts) atomicOr(err, TG_ERR_PAST_MEMORY);
    m.right_ts[id] = mts;
    if (m.right_active) m.right_active[id] = 1;
    atomicAnd((unsigned long long*)(m.has_msg + (id >> 6)), ~(1ull << (id & 63)));
  }
}

__device__ __forceinline__ void wb_step5(const tg_model& m, int64_t B, const int64_t* __restrict__ src,
                                         const int64_t* __restrict__ dst, const float* __restrict__ ts,
                                         const int64_t* __restrict__ eids, int64_t own, int64_t idx, uint32_t* err,
                                         int lane) {
  const int d4 = m.d / 4, e4 = m.d_e / 4;
  const int row4 = 3 * d4 + e4;
  const float* mem_ts = (m.msg_src == TG_SRC_LEFT) ? m.left_ts : m.right_ts;
  const float4* mem = reinterpret_cast<const float4*>((m.msg_src == TG_SRC_LEFT) ? m.left_vals : m.right_vals);
  const float4* nf = reinterpret_cast<const float4*>(m.nfeats);
  const float4* ef = reinterpret_cast<const float4*>(m.efeats);
  const float4* fq = reinterpret_cast<const float4*>(m.te_freq);
  const float4* ph = reinterpret_cast<const float4*>(m.te_phase);
  float4* box = reinterpret_cast<float4*>(m.msg_vals);
  const int64_t e = idx < B ? idx : idx - B;
  const int64_t own_id = idx < B ? src[e] : dst[e], other_id = idx < B ? dst[e] : src[e];  // node ids (features)
  const int64_t other = state_row(m, other_id);
  const float t = ts[e];
  const float dt = t - mem_ts[own];
  const int64_t eid = eids[e];
  for (int c = lane; c < row4; c += TG_WAVE) {
    float4 v;
    if (c < 2 * d4) {
      const int64_t node = c < d4 ? own : other;
      const int cc = c < d4 ? c : c - d4;
      v = mem[node * d4 + cc];
      if (nf) {
        const float4 f = nf[(c < d4 ? own_id : other_id) * d4 + cc];
        v.x += f.x; v.y += f.y; v.z += f.z; v.w += f.w;
      }
    } else if (c < 2 * d4 + e4) {
      v = ef ? ef[eid * e4 + (c - 2 * d4)] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const int cc = c - 2 * d4 - e4;
      const float4 w = fq[cc], q = ph[cc];
      v = make_float4(time_enc(dt, w.x, q.x), time_enc(dt, w.y, q.y), time_enc(dt, w.z, q.z), time_enc(dt, w.w, q.w));
    }
    box[own * row4 + c] = v;
  }
  if (lane == 0) {
    const uint64_t bit = 1ull << (own & 63);
    const unsigned long long old = atomicOr((unsigned long long*)(m.has_msg + (own >> 6)), bit);
    if (old & bit) atomicOr(err, TG_ERR_UNUSED_MESSAGE);
    m.msg_ts[own] = t;
  }
}

template <bool SYS = false>  // SYS: h is a window other GPUs store into (tg_part.h: system-scope loads)
__device__ __forceinline__ void wb_step6(const tg_model& m, int64_t id, int64_t idx, int64_t hrow,
                                         const float4* h, const float* __restrict__ ts, uint32_t* err,
                                         int lane) {
  const int w4 = m.d / 4;
  float4* left = reinterpret_cast<float4*>(m.left_vals);
  for (int c = lane; c < w4; c += TG_WAVE) left[id * w4 + c] = SYS ? ld_sys(h + hrow * w4 + c) : h[hrow * w4 + c];
  if (lane == 0) {
    const float nt = ts[idx];
    if (m.left_ts[id] > nt) atomicOr(err, TG_ERR_PAST_MEMORY);
    m.left_ts[id] = nt;
    if (m.left_active) m.left_active[id] = 1;
  }
}

// workgroup `bid` of `nblk`; rows_hi (nullable): the rows of index >= hi_from of a.left_row live in a peer-written window
// (tg_part_step: the push inbox) and are read there with system-scope loads
template <int PHASE>
__device__ __forceinline__ void writeback_body(const tg_model& m, const WritebackArgs& a, unsigned bid, unsigned nblk,
                                               const float4* rows_hi = nullptr, int64_t hi_from = 0) {
  const int lane = lane_id();
  const int64_t B = a.B;
  const int64_t n = min((int64_t)*a.n_upos, 2 * B);
  const int64_t wave0 = (int64_t)bid * 4 + (threadIdx.x >> 6), nwave = (int64_t)nblk * 4;
  const bool left_src = m.msg_src == TG_SRC_LEFT;
  const bool do5 = (PHASE == 0) ? left_src : !left_src;
  if (do5) {  // tiger.py:437-438 over all 2B positions
    const float* mem_ts = left_src ? m.left_ts : m.right_ts;
    for (int64_t i = wave0 * TG_WAVE + lane; i < 2 * B; i += nwave * TG_WAVE) {
      const int64_t e = i < B ? i : i - B;
      const int64_t node = i < B ? a.src[e] : a.dst[e];
      if (a.owner && a.owner[node] != a.my_rank) continue;  // another rank's node: its time is not kept here
      if (mem_ts[state_row(m, node)] > a.ts[e]) atomicOr(a.err, TG_ERR_EVENT_BEFORE_MEM);
    }
  }
  const int64_t po = a.plan_off ? 2 * *a.plan_off : 0;
  for (int64_t p = wave0; p < n; p += nwave) {
    const int64_t node_id = a.upos[p], idx = a.index[p];
    if (a.owner && a.owner[node_id] != a.my_rank) continue;  // partitioned state: the owner writes (wave-uniform)
    const int64_t id = state_row(m, node_id);
    if (PHASE == 0) {
      if (a.new_from_pending)
        wb_step4(m, id, id, reinterpret_cast<const float4*>(m.pending_vals), a.err, lane);
      else if (a.rows)
        wb_step4(m, id, a.new_row[po + idx], reinterpret_cast<const float4*>(a.rows), a.err, lane);
      else
        wb_step4(m, id, (int64_t)bm_rank(a.bm, a.rank, node_id), reinterpret_cast<const float4*>(a.reprs), a.err, lane);
    }
    if (do5) wb_step5(m, B, a.src, a.dst, a.ts, a.eids, id, idx, a.err, lane);
    if (PHASE == 1) {
      if (a.rows) {
        const int64_t hr = a.left_row[po + idx];
        if (rows_hi && hr >= hi_from) wb_step6<true>(m, id, idx, hr - hi_from, rows_hi, a.ts, a.err, lane);
        else wb_step6(m, id, idx, hr, reinterpret_cast<const float4*>(a.rows), a.ts, a.err, lane);
      }
      else
        wb_step6(m, id, idx, idx, reinterpret_cast<const float4*>(a.h), a.ts, a.err, lane);
    }
  }
  if (PHASE == 1) {
    if (a.clean_flags) {  // leave the step workspace zeroed for the next step (saves its memset launch)
      const int64_t tid = (int64_t)bid * blockDim.x + threadIdx.x, nth = (int64_t)nblk * blockDim.x;
      uint4* f = reinterpret_cast<uint4*>(a.clean_flags);
      for (int64_t i = tid; i < a.flag_bytes / 16; i += nth) f[i] = make_uint4(0u, 0u, 0u, 0u);
      const int64_t nb = a.clean_counts[0];  // involved count: only ranks below it were touched
      for (int64_t i = tid; i < nb; i += nth) a.clean_best[i] = 0ull;
    }
    // counts need no reset: the compaction overwrites [0] and [1], k_pos_max zeroes [2]
    if (bid == 0 && threadIdx.x == 0) {
      if (a.counts_dst)
        for (int i = 0; i < 4; ++i) a.counts_dst[i] = a.counts_src[i];
      if (a.clean_counts) a.clean_counts[3] = a.clean_counts[4] = 0;  // restarted-node count, batch-min-time key
      if (a.offset_dev) *a.offset_dev += B;
      if (a.lazy_batch) *a.lazy_batch += 1;
    }
  }
}


// ---- STEP 4-6 in ONE pass (eager updates, direct form), as a device function so that it can also RIDE on another launch.
// The hazard that forces two launches (tg_memory.hip) is STEP 5 reading message-memory rows that STEP 4 / STEP 6 of other
// wavefronts write.  Every row STEP 5 reads belongs to a positive node of the batch, so the launch that reads the centres
// takes a copy of exactly those 2B rows (a.snap: row + node features, a.snap_ts) before anything is written, and STEP 5
// builds the message from the copy:
//   mailbox[own] = [snap[own pos] | snap[other pos] | efeat | TE(t - snap_ts[own pos])]
// (msg_src = right: the copy is the right memory as STEP 4 leaves it, pending-or-right, as the reference reads it).
// Workgroup `bid` of `nblk`, any number of whole wavefronts per workgroup; one wavefront per unique positive node.
// ROWS6 = false: the VALUES of STEP 6 (left[v] <- h(t-), tiger.py:253-255) are written by somebody else - the epilogue of
// the product that computes h (GemmArgs.c2) - and only its bookkeeping (time, flags, invariant) happens here; nothing
// else in this pass depends on h, which is what lets it share that product's launch (WbRider).
template <bool ROWS6>
__device__ __forceinline__ void writeback_fused_body(const tg_model& m, const WritebackArgs& a, unsigned bid, unsigned nblk) {
  const int lane = lane_id();
  const int wpb = blockDim.x / TG_WAVE;
  const int64_t B = a.B;
  const int64_t n = min((int64_t)*a.n_upos, 2 * B);
  const int64_t wave0 = (int64_t)bid * wpb + (threadIdx.x / TG_WAVE), nwave = (int64_t)nblk * wpb;
  for (int64_t i = wave0 * TG_WAVE + lane; i < 2 * B; i += nwave * TG_WAVE) {  // tiger.py:437-438 over all 2B positions
    const int64_t e = i < B ? i : i - B;
    if (a.snap_ts[i] > a.ts[e]) atomicOr(a.err, TG_ERR_EVENT_BEFORE_MEM);
  }
  const int d4 = m.d / 4, e4 = m.d_e / 4;
  const int row4 = 3 * d4 + e4;
  const float4* snap = reinterpret_cast<const float4*>(a.snap);
  const float4* ef = reinterpret_cast<const float4*>(m.efeats);
  const float4* fq = reinterpret_cast<const float4*>(m.te_freq);
  const float4* ph = reinterpret_cast<const float4*>(m.te_phase);
  float4* box = reinterpret_cast<float4*>(m.msg_vals);
  const float4* pend = reinterpret_cast<const float4*>(m.pending_vals);
  const float4* hrow = reinterpret_cast<const float4*>(a.h);
  float4* right = reinterpret_cast<float4*>(m.right_vals);
  float4* left = reinterpret_cast<float4*>(m.left_vals);
  for (int64_t p = wave0; p < n; p += nwave) {
    // every load of the three steps is independent of every store: request them together (one wavefront has nothing
    // else to hide a dependent chain of five row fetches behind), then write
    const int64_t id = a.upos[p], idx = a.index[p];
    const int64_t e = idx < B ? idx : idx - B;
    const int64_t other_pos = idx < B ? B + e : e;
    const bool consume = bm_test(m.has_msg, id);  // STEP 4 applies (wave-uniform)
    const float t = a.ts[e];
    const float own_ts = a.snap_ts[idx];
    const int64_t eid = a.eids[e];
    const float mts = m.msg_ts[id], rts = m.right_ts[id], lts = m.left_ts[id];
    for (int c0 = 0; c0 < row4; c0 += TG_WAVE) {
      const int c = c0 + lane;
      float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), hv = pv, v = pv;
      if (c < d4) {
        if (consume) pv = pend[id * d4 + c];
        if (ROWS6) hv = hrow[idx * d4 + c];
        v = snap[idx * d4 + c];
      } else if (c < 2 * d4) {
        v = snap[other_pos * d4 + (c - d4)];
      } else if (c < 2 * d4 + e4) {
        if (ef) v = ef[eid * e4 + (c - 2 * d4)];
      } else if (c < row4) {
        const int cc = c - 2 * d4 - e4;
        const float4 w = fq[cc], q = ph[cc];
        const float dt = t - own_ts;
        v = make_float4(time_enc(dt, w.x, q.x), time_enc(dt, w.y, q.y), time_enc(dt, w.z, q.z), time_enc(dt, w.w, q.w));
      }
      if (c < d4) {
        if (consume) right[id * d4 + c] = pv;    // STEP 4: right <- pending (tiger.py:236-241)
        if (ROWS6) left[id * d4 + c] = hv;       // STEP 6: left <- h(t-)  (tiger.py:253-255)
      }
      if (c < row4) box[id * row4 + c] = v;      // STEP 5: [own | other | edge | time] (memory.py:89-106)
    }
    if (lane == 0) {
      if (consume) {
        if (rts > mts) atomicOr(a.err, TG_ERR_PAST_MEMORY);
        m.right_ts[id] = mts;
        if (m.right_active) m.right_active[id] = 1;
      }
      const uint64_t bit = 1ull << (id & 63);  // consumed (if it was set) and set again by the new message: stays / becomes set
      if (!consume) atomicOr((unsigned long long*)(m.has_msg + (id >> 6)), bit);
      m.msg_ts[id] = t;
      const float nt = a.ts[idx];
      if (lts > nt) atomicOr(a.err, TG_ERR_PAST_MEMORY);
      m.left_ts[id] = nt;
      if (m.left_active) m.left_active[id] = 1;
    }
  }
  {  // leave the step workspace zeroed for the next step
    const int64_t tid = (int64_t)bid * blockDim.x + threadIdx.x, nth = (int64_t)nblk * blockDim.x;
    if (a.clean_flags) {
      uint4* f = reinterpret_cast<uint4*>(a.clean_flags);
      for (int64_t i = tid; i < a.flag_bytes / 16; i += nth) f[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (a.clean_best && a.clean_best_by_pos) {
      for (int64_t i = tid; i < 2 * B; i += nth) a.clean_best[i < B ? a.src[i] : a.dst[i - B]] = 0ull;
    } else if (a.clean_best) {
      const int64_t nb = a.clean_counts[0];
      for (int64_t i = tid; i < nb; i += nth) a.clean_best[i] = 0ull;
    }
  }
  if (bid == 0 && threadIdx.x == 0) {
    if (a.counts_dst) {
      for (int i = 0; i < 4; ++i) a.counts_dst[i] = a.counts_src[i];
      if (a.clean_best_by_pos) a.counts_dst[0] = a.counts_dst[1] = -1;  // lean step: the sets were not formed
    }
    if (a.clean_counts) {
      a.clean_counts[3] = a.clean_counts[4] = 0;
      // the updater and the query-row product that follow read the number of unique positive nodes from here: slot 2
      // itself is reset by the first dedup pass of the NEXT batch, which may share a launch with them (collate prefetch)
      a.clean_counts[5] = *a.n_upos;
    }
    if (a.offset_dev) *a.offset_dev += B;
    if (a.lazy_batch) *a.lazy_batch += 1;
  }
}

// Write-back rider.  STEP 4 (right <- pending), STEP 5 (the new raw messages, built from the pre-batch snapshot) and the
// bookkeeping of STEP 6 depend on nothing the attention block computes; only the VALUES of STEP 6 are h(t-).  So the pass
// above (ROWS6 = false) runs as the FIRST `blocks` workgroups of the launch of the block's last product (fc2), on CUs that
// product leaves idle at C2 sizes and beside its matrix work at large sizes, and that product's epilogue stores the rows
// of the winning positions a second time, into the left memory (GemmArgs.c2 / c2_rows = PosArgs.win_row): the step has no
// write-back launch at all.  Hazards: nothing in the product's launch reads what the rider writes (right memory, mailbox,
// has-message bits, times) or what the epilogue scatters (left memory: STEP 5 reads the snapshot, not the table).
struct WbRider {
  tg_model m;
  WritebackArgs a;
  unsigned blocks;  // set by the launcher (a multiple of 8: the XCD map of the product's own blocks is unchanged)
  unsigned last;    // riders behind the product's blocks in the grid (else in front of them); set by the launcher
  // planned0 != 0 (tg_part_step): the rider is the FIRST launch of the planned, owner-filtered two-launch write-back (STEP 4 + 5,
  // or STEP 4 alone with msg_src = right: it reads no h(t-)); the product that hosts it stores no second copy of its rows
  int planned0;
  __device__ __forceinline__ void run(unsigned bid) const {
    if (planned0) writeback_body<0>(m, a, bid, blocks);
    else writeback_fused_body<false>(m, a, bid, blocks);
  }
};

}  // namespace tg
