// k_attn_tile: the temporal graph attention of STEP 3 (tiger/model/temporal_agg_modules.py:48-81,210-235 with the
// pre-multiplied weights of tg_attn_fuse) for a tile of 16 centres in ONE workgroup:
//
//   P0  centre rows c [16, d]                      global -> LDS
//   P1  G = c Wqk^T + gconst           [16, nk]    v_mfma_f32_16x16x4_f32, weights streamed fragment-major, G -> LDS
//   P2  per centre: gather the K neighbour rows, per-head scores against G, online softmax, weighted raw-row sum S;
//       G is read from LDS and S written over it (one wavefront per centre)
//   P3  t = relu([S | c] W1f^T + b1 + valid c1)    [16, d]   A operand straight from the LDS tiles
//   P4  h = t W2^T + b2                -> global
//
// G and S (2 x Q x nk floats per batch: 1.6 GB at the C5 shape) never touch HBM, and what were four launches
// (k_gemm_astat, k_attn_core, k_gemm_sk, k_gemm) is one.  Parallelism is over centres only, so a tile is 16 rows - the
// smallest MFMA tile - and a workgroup streams every weight once: the price is weight traffic from L2 (C2: 1.7 MB per
// workgroup), the gain is no intermediate round trips, no launch boundaries and no prologue / epilogue per product.
//
// LDS tiles are [16][ld] float with ld a multiple of 64 and the float4 chunks of row r XOR-permuted by r: the MFMA A
// fragment read (lane (r = l % 16, kg = l / 16) reads the float4 at k = 16 kc + 4 kg of row r: ds_read_b128, sixteen
// rows at a time) and the core's row accesses (one wavefront walks ONE row) are both conflict free.
#include "tg_step.h"
#include "tg_tile.h"

namespace tg {

typedef float f32x4t __attribute__((ext_vector_type(4)));

struct TileArgs {
  int64_t Q;
  const float* cc;  // [Q, d] centre rows (memory row + node features), written by the centres launch
  const float* ts;  // [Q]
  const int64_t* l1_nids;
  const int64_t* l1_eids;
  const float* l1_ts;
  const float* reprs;  // compact form (direct == 0): rows of the involved nodes, addressed through (bm, rank)
  const uint64_t* bm;
  const uint32_t* rank;
  int direct;          // eager updates, direct form: neighbour rows are read from pending / right by node id
  const float* tile;   // tile section of the fused blob (tg_tile.h)
  const float* zeros;  // zero_line()
  float* out;          // [Q, d]
  TileDims t;
  PosArgs pos;
  int dbg;  // diagnostic (TG_TILE_DBG=1): s_memtime stamps of every wavefront at the phase boundaries, 0 in production
};

// diagnostic only: [workgroup < 512][wavefront < 16][8] stamps {entry, P0 done, P1 done, P2 own centre done, P2 barrier, P3 done, P4 done}
__device__ unsigned long long g_tile_trace[512 * 16 * 8];

__device__ __forceinline__ float4 ldg4t(const float* p) { return *reinterpret_cast<const float4*>(p); }
// float offset of (row, col) inside a swizzled tile
__device__ __forceinline__ int swz(int row, int col, int ld) { return row * ld + ((((col >> 2) ^ row)) << 2) + (col & 3); }

template <int W>
struct RV {
  float a[W];
};
// The load itself is UNCONDITIONAL and its result is used as it is: lanes past the end of the row read the zero line.
template <int W>
__device__ __forceinline__ RV<W> gld(const float* __restrict__ row, int col, int width, const float* __restrict__ zl) {
  RV<W> r;
  const float* p = col < width ? row + col : zl;
  if (W == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    r.a[0] = v.x; r.a[1] = v.y; r.a[2] = v.z; r.a[W - 1] = v.w;
  } else {
    const float2 v = *reinterpret_cast<const float2*>(p);
    r.a[0] = v.x; r.a[W - 1] = v.y;
  }
  return r;
}
template <int W>
__device__ __forceinline__ RV<W> lld(const float* S, int row, int ld, int col, bool in) {
  RV<W> r;
  const float* p = S + swz(row, in ? col : 0, ld);
  if (W == 4) {
    float4 v = *reinterpret_cast<const float4*>(p);
    if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
    r.a[0] = v.x; r.a[1] = v.y; r.a[2] = v.z; r.a[W - 1] = v.w;
  } else {
    float2 v = *reinterpret_cast<const float2*>(p);
    if (!in) v = make_float2(0.f, 0.f);
    r.a[0] = v.x; r.a[W - 1] = v.y;
  }
  return r;
}
template <int W>
__device__ __forceinline__ void lst(float* S, int row, int ld, int col, bool in, const RV<W>& r) {
  if (!in) return;
  float* p = S + swz(row, col, ld);
  if (W == 4) *reinterpret_cast<float4*>(p) = make_float4(r.a[0], r.a[1], r.a[2], r.a[W - 1]);
  else *reinterpret_cast<float2*>(p) = make_float2(r.a[0], r.a[W - 1]);
}

// ---- one run of MFMA steps: G column tiles nt[0..G) over k-chunks [kc0, kc1); a_of(kc) = this lane's A float4 ----
// The weight blocks of a run are contiguous in memory (chunk-major inside a column tile) and come from L2, ~2.5 k cycles
// away, while a chunk is 4 G MFMAs of work: the blocks travel PF chunks ahead in a ring of register slots (first version,
// one chunk ahead: the fc1 phase ran at 45 % of the matrix rate, every chunk waiting for its block).  The loop body has
// no guards (addresses are clamped to the last chunk instead), so the wait counts stay exact; the < PF chunks that are
// left over are in the ring already.
template <int G, int PF, class AFn>
__device__ __forceinline__ void mm_run(const float* __restrict__ wf, int KC, const int (&nt)[G], int kc0, int kc1, AFn&& a_of,
                                       f32x4t (&acc)[G], int lane) {
  const float* wp[G];
#pragma unroll
  for (int g = 0; g < G; ++g) wp[g] = wf + (size_t)nt[g] * KC * 256 + lane * 4;
  const int last = kc1 - 1;
  float4 b[PF][G];
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int kk = min(kc0 + s, last);
#pragma unroll
    for (int g = 0; g < G; ++g) b[s][g] = ldg4t(wp[g] + (size_t)kk * 256);
  }
  float4 a = a_of(kc0);
  auto steps = [&](const float4& av, const float4 (&bv)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[g].x, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[g].y, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[g].z, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[g].w, acc[g], 0, 0, 0);
  };
  int kc = kc0;
  for (; kc + PF <= kc1; kc += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const float4 an = a_of(min(kc + s + 1, last));
      float4 cur[G];
#pragma unroll
      for (int g = 0; g < G; ++g) cur[g] = b[s][g];
      const int kk = min(kc + s + PF, last);
#pragma unroll
      for (int g = 0; g < G; ++g) b[s][g] = ldg4t(wp[g] + (size_t)kk * 256);
      steps(a, cur);
      a = an;
    }
  }
#pragma unroll
  for (int s = 0; s < PF - 1; ++s) {
    if (kc + s < kc1) {
      const float4 an = a_of(min(kc + s + 1, last));
      steps(a, b[s]);
      a = an;
    }
  }
}

template <int G, int NWV, class AFn, class Epi>
__device__ __forceinline__ void mm_whole(const float* __restrict__ wf, int KC, int first, AFn&& a_of, Epi&& epi, int wave,
                                         int lane) {
  int nt[G];
  f32x4t acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    nt[g] = wave + NWV * (first + g);
    acc[g] = f32x4t{0.f, 0.f, 0.f, 0.f};
  }
  mm_run<G, (G == 4 ? 3 : (G == 2 ? 4 : 8))>(wf, KC, nt, 0, KC, a_of, acc, lane);
#pragma unroll
  for (int g = 0; g < G; ++g) epi(nt[g], acc[g]);
}

// One product of the tile: out[16, 16 NT] = A[16, 16 KC] W^T.  The first (NT / NWV) * NWV column tiles are dealt whole,
// round robin, up to four per wavefront at a time (one A fragment read feeds them all); the (column tile, k-chunk)
// units of the remaining R < NWV tiles are dealt evenly - a wavefront's run touches at most two tiles - and their
// partial sums are folded through LDS in wavefront order (a fixed order: the result does not depend on timing).
// Ends with a barrier: what `epi` wrote is visible to the next phase.
template <int NWV, class AFn, class Epi>
__device__ __forceinline__ void mm_phase(const float* __restrict__ wf, int NT, int KC, AFn&& a_of, Epi&& epi,
                                         float* __restrict__ red, int wave, int lane) {
  const int per = NT / NWV, NTw = per * NWV;
  int i = 0;
  for (; i + 4 <= per; i += 4) mm_whole<4, NWV>(wf, KC, i, a_of, epi, wave, lane);
  for (; i + 2 <= per; i += 2) mm_whole<2, NWV>(wf, KC, i, a_of, epi, wave, lane);
  for (; i < per; ++i) mm_whole<1, NWV>(wf, KC, i, a_of, epi, wave, lane);
  const int R = NT - NTw;
  if (R > 0) {  // uniform over the workgroup
    const int units = R * KC, U = (units + NWV - 1) / NWV;
    int u = wave * U;
    const int u1 = min(units, u + U);
    int slot = 0;
    while (u < u1) {
      const int r = u / KC, kc0 = u - r * KC, kc1 = min(KC, kc0 + (u1 - u));
      const int nt[1] = {NTw + r};
      f32x4t acc[1] = {f32x4t{0.f, 0.f, 0.f, 0.f}};
      mm_run<1, 8>(wf, KC, nt, kc0, kc1, a_of, acc, lane);
      float* p = red + (wave * 2 + slot) * 256 + lane;
      p[0] = acc[0][0]; p[64] = acc[0][1]; p[128] = acc[0][2]; p[192] = acc[0][3];
      ++slot;
      u += kc1 - kc0;
    }
    __syncthreads();
    for (int r = wave; r < R; r += NWV) {
      const int w0 = (r * KC) / U, w1 = ((r + 1) * KC - 1) / U;
      f32x4t s = f32x4t{0.f, 0.f, 0.f, 0.f};
      for (int ww = w0; ww <= w1; ++ww) {
        const int sl = ((ww * U) / KC == r) ? 0 : 1;  // slot 0 holds the first tile a wavefront's run touches
        const float* p = red + (ww * 2 + sl) * 256 + lane;
        s[0] += p[0]; s[1] += p[64]; s[2] += p[128]; s[3] += p[192];
      }
      epi(NTw + r, s);
    }
  }
  __syncthreads();
}

// ---- P2: one centre.  The body of k_attn_core (tg_model.hip) with G read from / S written to the LDS tile. ----
template <int NH, int W>
__device__ __forceinline__ bool core_centre(const tg_model& m, const TileArgs& a, int64_t i, int row, float* __restrict__ GS,
                                            int64_t nb_l, int64_t eid_l, float dt_l, int u_l, const RV<W>& w4,
                                            const RV<W>& p4, int lane) {
  using V = RV<W>;
  const int d = a.t.d, de = a.t.de, kvw = a.t.kvw, ld = a.t.gs_ld;
  const int c = lane * W;
  const bool in_d = c < d, in_e = c < de;
  const unsigned long long live = __ballot(nb_l != 0);  // padding keys are masked (temporal_agg_modules.py:80)
  const bool any = live != 0ull;
  // The folded query g stays in LDS and is re-read for every key (6 conflict-free reads of this wavefront's own row):
  // held in registers as in k_attn_core it pushed the kernel over the 168 VGPRs of three wavefronts per SIMD, and the
  // spill reloads inside the key loop (scratch loads are counted with the row gathers) serialised the gathers.
  V acc[NH][3];
  float mx[NH], l[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    mx[h] = -INFINITY;
    l[h] = 0.f;
#pragma unroll
    for (int j = 0; j < W; ++j) acc[h][0].a[j] = acc[h][1].a[j] = acc[h][2].a[j] = 0.f;
  }
  constexpr int PD = W == 2 ? 4 : 3;  // raw rows of the next keys in flight while the current key is reduced
  V ya[PD], yn[PD], yb[PD];
  auto fetch = [&](int slot, int k) {
    const int64_t u = __shfl(u_l, k, TG_WAVE);
    const int64_t nb = __shfl(nb_l, k, TG_WAVE);
    const int64_t eid = __shfl(eid_l, k, TG_WAVE);
    const float* nrow = a.direct ? ((u & 1) ? m.pending_vals : m.right_vals) + (u >> 1) * d : a.reprs + u * d;
    const int wd = (a.dbg & 4) ? 0 : d, we = (a.dbg & 4) ? 0 : de;  // diagnostic: bit 4 = no gathers (every lane reads the zero line)
    ya[slot] = gld<W>(nrow, c, wd, a.zeros);
    yn[slot] = gld<W>(m.nfeats ? m.nfeats + nb * d : nrow, c, m.nfeats ? wd : 0, a.zeros);
    yb[slot] = gld<W>(m.efeats ? m.efeats + eid * de : nrow, c, m.efeats ? we : 0, a.zeros);
  };
  auto reduce = [&](int slot, int k) {
    const float dt = __shfl(dt_l, k, TG_WAVE);
    V x[3];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      x[0].a[j] = ya[slot].a[j] + yn[slot].a[j];
      x[1].a[j] = yb[slot].a[j];
      x[2].a[j] = c + j < d ? time_enc_fast(dt, w4.a[j], p4.a[j]) : 0.f;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      V g[3];
      g[0] = lld<W>(GS, row, ld, h * kvw + c, in_d);
      g[1] = lld<W>(GS, row, ld, h * kvw + d + c, in_e);
      g[2] = lld<W>(GS, row, ld, h * kvw + d + de + c, in_d);
      float p = 0.f;
#pragma unroll
      for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
        for (int j = 0; j < W; ++j) p = fmaf(g[sgm].a[j], x[sgm].a[j], p);
      p = wave_sum(p);  // wave-uniform
      float b = 1.f;
      if (p > mx[h]) {  // new running maximum: rescale what has been accumulated (uniform branch)
        const float s = expf(mx[h] - p);
        l[h] *= s;
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
          for (int j = 0; j < W; ++j) acc[h][sgm].a[j] *= s;
        mx[h] = p;
      } else {
        b = expf(p - mx[h]);
      }
      l[h] += b;
#pragma unroll
      for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[h][sgm].a[j] = fmaf(b, x[sgm].a[j], acc[h][sgm].a[j]);
    }
  };
  // Keys are walked in list order, padding included, PD at a time; key k travels in ring slot k % PD.  Every fetch is
  // unconditional (a padding key reads node / edge row 0, which exists and is never used) and only the arithmetic is
  // skipped for padding keys: a branch that holds vector-memory instructions makes the compiler's wait-count pass give
  // up at the join and wait for EVERYTHING in flight before the next key - the ring then hides nothing (the cursor
  // form of k_attn_core, with its guarded fetches, spends one full memory latency per key).
  const int K = m.n_neighbors;
#pragma unroll
  for (int sl = 0; sl < PD; ++sl) fetch(sl, min(sl, K - 1));
  for (int k0 = 0; k0 < K; k0 += PD) {
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) {
      const int k = k0 + sl;
      if (k < K && ((live >> k) & 1ull) && !(a.dbg & 2)) reduce(sl, k);  // diagnostic: bit 2 = no arithmetic
      fetch(sl, min(k + PD, K - 1));
    }
  }
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const float inv = any ? 1.f / l[h] : 0.f;
#pragma unroll
    for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
      for (int j = 0; j < W; ++j) acc[h][sgm].a[j] *= inv;
    lst<W>(GS, row, ld, h * kvw + c, in_d, acc[h][0]);
    lst<W>(GS, row, ld, h * kvw + d + c, in_e, acc[h][1]);
    lst<W>(GS, row, ld, h * kvw + d + de + c, in_d, acc[h][2]);
  }
  return any;
}

// NWV wavefronts per workgroup, CPW centres per wavefront in P2: a tile holds MC = NWV * CPW <= 16 centres (rows MC..15 of
// the MFMA tiles are padding).  (12, 1): 168 VGPRs per lane keep the core free of spills at three wavefronts per SIMD and
// a C2-sized batch (3 072 centres) is exactly one tile per CU; (8, 2): full 16-row tiles for launches of many rounds,
// where the matrix work per centre is what counts.  (16, 1) would cap the core at 128 VGPRs: 129 spilled.
template <int NH, int W, int NWV, int CPW>
__global__ void __launch_bounds__(64 * NWV) k_attn_tile(tg_model m, TileArgs a) {
  constexpr int MC = NWV * CPW;
  static_assert(MC <= TILE_M, "a tile is one 16-row MFMA tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const TileDims& t = a.t;
  float* GS = lds;
  float* CT = GS + TILE_M * t.gs_ld;
  float* TT = CT + TILE_M * t.c_ld;
  float* RED = TT + TILE_M * t.c_ld;
  int* valid_s = reinterpret_cast<int*>(RED + NWV * 2 * 256);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.pos.advance_off && blockIdx.x == 0 && tid == 0) *a.pos.advance_off += a.pos.B;  // lean embed-only step
  const int d = t.d, K = m.n_neighbors;
  const float* wqk = a.tile + t.o_wqk;
  const float* gconst = a.tile + t.o_gconst;
  const float* w1f = a.tile + t.o_w1f;
  const float* b1 = a.tile + t.o_b1;
  const float* c1 = a.tile + t.o_c1;
  const float* w2 = a.tile + t.o_w2;
  const float* b2 = a.tile + t.o_b2;
  const int fr = lane & 15, fk = lane >> 4;  // MFMA fragment coordinates: A row / B column, k group
  RV<W> w4 = gld<W>(m.te_freq, lane * W, d, a.zeros), p4 = gld<W>(m.te_phase, lane * W, d, a.zeros);
  const int64_t ntiles = (a.Q + MC - 1) / MC;
  const bool trace = a.dbg && blockIdx.x < 512 && lane == 0;
  unsigned long long* tr = g_tile_trace + ((size_t)blockIdx.x * 16 + wave) * 8;
  if (trace) tr[0] = __builtin_amdgcn_s_memtime();
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t i0 = tile * MC;
    // ---- per-key metadata of this wavefront's centres, one key per lane: requested now, needed in P2
    int64_t nb_l[CPW], eid_l[CPW];
    float dt_l[CPW];
    int u_l[CPW];
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
      const int64_t i = i0 + wave + NWV * s;
      nb_l[s] = 0; eid_l[s] = 0; dt_l[s] = 0.f; u_l[s] = 0;
      if (lane < K && i < a.Q) {
        nb_l[s] = a.l1_nids[i * K + lane];
        eid_l[s] = a.l1_eids[i * K + lane];
        dt_l[s] = a.ts[i] - a.l1_ts[i * K + lane];
      }
    }
    // ---- P0: centre rows -> LDS (zeros past d and past Q)
    {
      const int c4 = t.c_ld >> 2;
      for (int e = tid; e < TILE_M * c4; e += 64 * NWV) {
        const int r = e / c4, col = (e - r * c4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < d && r < MC && i0 + r < a.Q) v = ldg4t(a.cc + (i0 + r) * d + col);
        *reinterpret_cast<float4*>(CT + swz(r, col, t.c_ld)) = v;
      }
    }
    __syncthreads();
    // the loads that depend on the neighbour ids (has-message bit or rank; the time invariants of a lean step) are
    // requested here: they are in flight while P1 runs
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
      if (nb_l[s] != 0) {
        u_l[s] = a.direct ? (int)(2 * nb_l[s] + (bm_test(m.has_msg, nb_l[s]) ? 1 : 0)) : (int)bm_rank(a.bm, a.rank, nb_l[s]);
        if (a.direct && a.pos.chk_err && (u_l[s] & 1)) check_msg_times(m, nb_l[s], a.pos.chk_err);
      }
    }
    if (trace) tr[1] = __builtin_amdgcn_s_memtime();
    // ---- P1: G = c Wqk^T + gconst
    mm_phase<NWV>(
        wqk, t.NTg, t.KCd,
        [&](int kc) { return *reinterpret_cast<const float4*>(CT + swz(fr, 16 * kc + 4 * fk, t.c_ld)); },
        [&](int nt, const f32x4t& acc) {
          const int col = nt * 16 + fr;
          const float gc = gconst[col];
#pragma unroll
          for (int j = 0; j < 4; ++j) GS[swz(4 * fk + j, col, t.gs_ld)] = acc[j] + gc;
        },
        RED, wave, lane);
    if (trace) tr[2] = __builtin_amdgcn_s_memtime();
    // ---- P2: gather + scores + softmax + weighted raw-row sum, S over G
#pragma unroll
    for (int s = 0; s < CPW; ++s) {
      const int row = wave + NWV * s;
      const int64_t i = i0 + row;
      bool any = false;
      if (i < a.Q) any = core_centre<NH, W>(m, a, i, row, GS, nb_l[s], eid_l[s], dt_l[s], u_l[s], w4, p4, lane);
      if (lane == 0) valid_s[row] = any ? 1 : 0;
    }
    if (MC < TILE_M && tid >= MC && tid < TILE_M) valid_s[tid] = 0;  // padding rows
    if (trace) tr[3] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (trace) tr[4] = __builtin_amdgcn_s_memtime();
    // ---- P3: t = relu([S | c] W1f^T + b1 + valid c1)
    mm_phase<NWV>(
        w1f, t.NTd, t.KCnk + t.KCd,
        [&](int kc) {
          const float* p = kc < t.KCnk ? GS + swz(fr, 16 * kc + 4 * fk, t.gs_ld)
                                       : CT + swz(fr, 16 * (kc - t.KCnk) + 4 * fk, t.c_ld);
          return *reinterpret_cast<const float4*>(p);
        },
        [&](int nt, const f32x4t& acc) {
          const int col = nt * 16 + fr;
          const float bb = b1[col], cb = c1[col];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * fk + j;
            const float v = acc[j] + bb + (valid_s[r] ? cb : 0.f);
            TT[swz(r, col, t.c_ld)] = fmaxf(v, 0.f);
          }
        },
        RED, wave, lane);
    if (trace) tr[5] = __builtin_amdgcn_s_memtime();
    // ---- P4: h = t W2^T + b2 -> global
    mm_phase<NWV>(
        w2, t.NTd, t.KCd,
        [&](int kc) { return *reinterpret_cast<const float4*>(TT + swz(fr, 16 * kc + 4 * fk, t.c_ld)); },
        [&](int nt, const f32x4t& acc) {
          const int col = nt * 16 + fr;
          if (col < d) {
            const float bb = b2[col];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * fk + j;
              if (r < MC && i0 + r < a.Q) a.out[(i0 + r) * d + col] = acc[j] + bb;
            }
          }
        },
        RED, wave, lane);
    if (trace) tr[6] = __builtin_amdgcn_s_memtime();
  }
  // what rode on the core launch: the second dedup pass of the step (a chain of four dependent memory round trips for
  // the few thousand positions of a batch).  Last of all and by one wavefront per workgroup only: nothing waits for it.
  if (a.pos.best && wave == NWV - 1)
    pos_winners_pass(a.pos, (int64_t)blockIdx.x * 64 + lane, (int64_t)gridDim.x * 64);
}

// dynamic LDS beyond the default limit has to be requested once per kernel
template <int NH, int W, int NWV, int CPW>
static int tile_attr() {
  // (the attribute is per device: set once for every device this process launches the kernel on)
  constexpr int MAXD = 64;
  static bool attr_set[MAXD] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXD) dev = -1;
  if (dev < 0 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_tile<NH, W, NWV, CPW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS_MAX);
    if (e != hipSuccess) {
      set_hip_error(e, "k_attn_tile: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
      return TG_EHIP;
    }
    if (dev >= 0) attr_set[dev] = true;
  }
  return TG_OK;
}
template <int NH, int W, int NWV, int CPW>
static int launch_tile(const tg_model* m, const TileArgs& a, size_t lds, hipStream_t st) {
  int rc;
  if ((rc = tile_attr<NH, W, NWV, CPW>()) != TG_OK) return rc;
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(a.Q, NWV * CPW), 256 * 64);
  hipLaunchKernelGGL((k_attn_tile<NH, W, NWV, CPW>), dim3(grid), dim3(64 * NWV), lds, st, *m, a);
  return check_launch("attn_tile");
}
// called by tg_attn_fuse (never inside a stream capture): the first launch may then happen inside one
int attn_tile_prepare() {
  int rc = TG_OK;
#define TG_TA(NH_, W_) \
  if (rc == TG_OK) rc = tile_attr<NH_, W_, 12, 1>(); \
  if (rc == TG_OK) rc = tile_attr<NH_, W_, 8, 2>()
  TG_TA(1, 4); TG_TA(1, 2); TG_TA(2, 4); TG_TA(2, 2); TG_TA(4, 4); TG_TA(4, 2);
#undef TG_TA
  return rc;
}

// 1 when attn_tile_launch would run for this model (the dimensions fit one workgroup's LDS and the knob is on)
int attn_tile_applies(const tg_model* m) {
  // OFF by default: measured on MI355X the one-launch form is slower than the four launches it replaces (C2: 97 us
  // against 76; C5 shape 3.98 ms against 2.97) - a 16-centre tile needs 32 B/clk of weights per CU at full matrix rate
  // and a CU pulls 13-16 B/clk of L2-resident lines here, while the 64-row tiles of the separate products need a quarter
  // of that and their G / S round trip costs less than it saves (profiles/r03_attn_tile_phase_trace.txt)
  static const int knob = getenv("TG_ATTN_TILE") ? atoi(getenv("TG_ATTN_TILE")) : 0;
  return (knob != 0 && tile_waves(m) != 0 && !m->row_of) ? 1 : 0;
}

int attn_tile_launch(const tg_model* m, int64_t Q, const float* cc, const float* ts, const int64_t* l1_nids,
                     const int64_t* l1_eids, const float* l1_ts, const float* reprs, const uint64_t* bm,
                     const uint32_t* rank, float* out, int direct, const PosArgs* pos, hipStream_t st) {
  if (!tile_waves(m)) return TG_EUNSUPPORTED;
  TileArgs a{};
  a.Q = Q; a.cc = cc; a.ts = ts; a.l1_nids = l1_nids; a.l1_eids = l1_eids; a.l1_ts = l1_ts;
  a.reprs = reprs; a.bm = bm; a.rank = rank; a.direct = direct; a.out = out;
  a.t = tile_dims(m);
  a.tile = m->attn_fused + attn_fused_floats_of((size_t)a.t.d, (size_t)a.t.nk);
  a.pos = pos ? *pos : PosArgs{};
  a.zeros = zero_line();
  if (!a.zeros) return TG_EHIP;
  static const int dbg_knob = getenv("TG_TILE_DBG") ? atoi(getenv("TG_TILE_DBG")) : 0;
  a.dbg = dbg_knob;
  // tiles of 12 centres (12 wavefronts) while they need no more rounds of 256 workgroups than tiles of 16 would: the
  // matrix work of a workgroup is that of a 16-row tile either way, so below that point more CUs share the centres
  static const int mc_knob = getenv("TG_ATTN_TILE_MC") ? atoi(getenv("TG_ATTN_TILE_MC")) : 0;
  const bool twelve = mc_knob ? mc_knob == 12 : cdiv(cdiv(Q, 12), 256) <= cdiv(cdiv(Q, 16), 256);
  const size_t lds = tile_lds_bytes(a.t, twelve ? 12 : 8);
  const int W = std::max(a.t.d, a.t.de) <= 128 ? 2 : 4;  // narrow rows: two columns per lane fill more lanes (as k_attn_core)
  const int nh = a.t.nh;
#define TG_TILE(NH_, W_)                                                  \
  if (nh == NH_ && W == W_)                                               \
    return twelve ? launch_tile<NH_, W_, 12, 1>(m, a, lds, st) : launch_tile<NH_, W_, 8, 2>(m, a, lds, st)
  TG_TILE(2, 4); TG_TILE(2, 2); TG_TILE(1, 4); TG_TILE(1, 2); TG_TILE(4, 4); TG_TILE(4, 2);
#undef TG_TILE
  return TG_EUNSUPPORTED;
}

}  // namespace tg

extern "C" int tg_debug_tile_trace(unsigned long long* out_host, int n_blocks) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(tg::g_tile_trace), sizeof(unsigned long long) * 16 * 8 * n_blocks) == hipSuccess ? 0 : -4;
}
