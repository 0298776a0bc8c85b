// STEP 1-3 of TIGE.contrast_learning on device: message consumption + memory update
// (tiger/model/tiger.py:208-221,292-356) and the one-layer temporal graph attention
// (tiger/model/temporal_agg_modules.py:29-83,186-235).  SURVEY.md K6, K8; a13-a18.
//
// Attention is restructured around the fact that there is ONE query per centre and K
// keys: instead of projecting every key/value row (24*K*d^2 flop per centre) the query
// is folded through Wk (g_h = Wk_h^T q_h), scores are plain dot products with the raw
// key rows, the softmax-weighted raw rows are summed first and projected through Wv
// once.  q.bk is constant over keys and cancels in the softmax; sum(a)=1 carries bv.
// Mathematically identical, ~5x fewer flops, and the K*3d key rows are touched once by
// a gather kernel instead of being materialised for a GEMM.
#include "tg_step.h"
#include "tg_sample.h"
#include "tg_tile.h"

namespace tg {

// ---- invariants of compute_messages (message_modules.py:158-159, tiger.py:325-327) ----
__global__ void k_check_messages(tg_model m, const int64_t* __restrict__ outdated, const int32_t* __restrict__ n_dev,
                                 int64_t cap, uint32_t* __restrict__ err) {
  const int64_t n = min((int64_t)*n_dev, cap);
  const float* mem_ts = (m.msg_src == TG_SRC_LEFT) ? m.left_ts : m.right_ts;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = outdated[i];
    const float mts = m.msg_ts[id], last = mem_ts[id];
    if (last > mts) atomicOr(err, TG_ERR_MSG_BEFORE_MEM);
    if (m.msg_src == TG_SRC_LEFT && !(mts == last)) atomicOr(err, TG_ERR_MSG_TS_MISMATCH);
  }
}

// centre rows: c_i = reprs[local(nid_i)] + nfeat[nid_i]   (temporal_agg_modules.py:48-50)
// The last `qblocks` blocks of the grid instead compute the constant half of the query
// projection, qconst[n] = bq[n] + sum_j Wq[n, d + j] * cos(phase[j])   (TE(0) = cos(phi)).
__global__ void __launch_bounds__(256) k_attn_centres(int64_t Q, int d4, const int64_t* __restrict__ nids,
                                                      const float4* __restrict__ reprs, const uint64_t* __restrict__ bm,
                                                      const uint32_t* __restrict__ rank, const float4* __restrict__ nf,
                                                      float4* __restrict__ out, int qblocks, const float* __restrict__ wq,
                                                      const float* __restrict__ bq, const float* __restrict__ freq,
                                                      const float* __restrict__ phase, float* __restrict__ qconst,
                                                      PosArgs pos) {
  const int cblocks = (int)gridDim.x - qblocks;
  if (pos.best && (int)blockIdx.x < cblocks)  // second dedup pass rides on the centre blocks
    pos_winners_pass(pos, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)cblocks * blockDim.x);
  if ((int)blockIdx.x >= cblocks) {
    const int d = d4 * 4, lane = lane_id();
    const int n = ((int)blockIdx.x - cblocks) * 4 + (threadIdx.x >> 6);
    if (n >= 2 * d) return;
    float acc = 0.f;
    for (int j = lane; j < d; j += TG_WAVE) acc += wq[(int64_t)n * 2 * d + d + j] * time_enc(0.f, freq[j], phase[j]);
    acc = wave_sum(acc);
    if (lane == 0) qconst[n] = acc + bq[n];
    return;
  }
  const int64_t total = Q * d4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)cblocks * blockDim.x) {
    const int64_t i = t / d4;
    const int c = (int)(t - i * d4);
    const int64_t id = nids[i];
    float4 v = reprs[(int64_t)bm_rank(bm, rank, id) * d4 + c];
    if (nf) {
      const float4 f = nf[id * d4 + c];
      v.x += f.x; v.y += f.y; v.z += f.z; v.w += f.w;
    }
    out[t] = v;
  }
}

__global__ void __launch_bounds__(256) k_attn_centres_direct(tg_model m, int64_t Q, const int64_t* __restrict__ nids,
                                                             const float4* __restrict__ nf, float4* __restrict__ out,
                                                             DirectArgs da, PosArgs pos) {
  centres_direct_body(m, Q, ArrayIds{nids, pos.ts}, nf, out, da, pos,
                      (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// explicit fma: the library is built with -ffp-contract=off (only the time encoding needs the
// unfused product), so contractions are spelled out where they are wanted
__device__ __forceinline__ float dot4(float4 a, float4 b, float acc) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}
__device__ __forceinline__ void axpy4(float4& s, float b, float4 x) {
  s.x = fmaf(b, x.x, s.x);
  s.y = fmaf(b, x.y, s.y);
  s.z = fmaf(b, x.z, s.z);
  s.w = fmaf(b, x.w, s.w);
}
__device__ __forceinline__ void scale4(float4& s, float a) {
  s.x *= a; s.y *= a; s.z *= a; s.w *= a;
}

// One wavefront per centre.  Streams its K neighbour rows once: node part
// reprs[local]+nfeat, edge part efeat, time part cos(dt*w+phi); per head an online
// softmax over the keys accumulates the weighted raw row.  A lane owns W consecutive columns
// of every segment (NV such groups, i.e. widths up to 64*W*NV); W is picked per model so that
// the 64 lanes are as full as possible: the kernel is VALU-bound (time encoding + 2*NH fmas per
// column and key), and d = 172 fills 43 lanes as float4 but 58 lanes as three floats.
// Latency structure: lane k first resolves key k's metadata (neighbour id -> local row via
// the rank popcount, edge id, dt) for all K keys at once, the per-key loop then only
// broadcasts it, and the raw rows of the next keys are requested before key k is reduced.
template <int W>
struct RowVec {
  float a[W];
};
struct __attribute__((packed, aligned(4))) F3 {
  float x, y, z;
};
// columns [col, col+W) of a row of `width` floats (row 16-byte aligned, width % 4 == 0); zeros past the end
template <int W>
__device__ __forceinline__ RowVec<W> row_load(const float* __restrict__ row, int col, int width, const float* __restrict__ zl) {
  RowVec<W> r;
  // W = 4, 2: the load is UNCONDITIONAL and its result is used as it is - lanes past the end of the row read the zero line zl.
  // (A load inside a divergent branch makes the compiler's wait-count pass wait for every load in flight at the join,
  // and so does a select on the loaded value scheduled right behind the load: either way the gathers of the key ring
  // were serialised, one full memory latency per key.)
  if (W == 4) {
    const float4 v = *reinterpret_cast<const float4*>(col < width ? row + col : zl);
    r.a[0] = v.x; r.a[1] = v.y; r.a[2] = v.z; r.a[W - 1] = v.w;
  } else if (W == 2) {  // rows are 16-byte aligned and widths multiples of 4: an 8-byte access never straddles the row end
    const float2 v = *reinterpret_cast<const float2*>(col < width ? row + col : zl);
    r.a[0] = v.x; r.a[W - 1] = v.y;
  } else if (W == 3) {
    if (col + 3 <= width) {
      const F3 v = *reinterpret_cast<const F3*>(row + col);
      r.a[0] = v.x; r.a[1] = v.y; r.a[W - 1] = v.z;
    } else {  // the lane that straddles the row end (one per wave) and the idle lanes
#pragma unroll
      for (int j = 0; j < W; ++j) r.a[j] = col + j < width ? row[col + j] : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < W; ++j) r.a[j] = col + j < width ? row[col + j] : 0.f;
  }
  return r;
}
template <int W>
__device__ __forceinline__ void row_store(float* __restrict__ row, int col, int width, const RowVec<W>& r) {
  if (W == 4) {
    // streaming (non-temporal) stores: the S rows - 12.7 MB per C2 batch, the step's largest output - are read once, by the
    // next launch, from other XCDs: kept out of this XCD's L2 they do not wait for its write-back at the end of the kernel
    // (C2, same box: core 17.86 -> 16.82 us, fc1 behind it 22.45 -> 23.02 us, step 85.40 -> 84.64 us)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f x = {r.a[0], r.a[1], r.a[2], r.a[W - 1]};
    if (col < width) __builtin_nontemporal_store(x, reinterpret_cast<v4f*>(row + col));
  } else if (W == 2) {
    if (col < width) *reinterpret_cast<float2*>(row + col) = make_float2(r.a[0], r.a[W - 1]);
  } else if (W == 3 && col + 3 <= width) {
    F3 v;
    v.x = r.a[0]; v.y = r.a[1]; v.z = r.a[W - 1];
    *reinterpret_cast<F3*>(row + col) = v;
  } else {
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (col + j < width) row[col + j] = r.a[j];
  }
}

// broadcast of lane k's pointer / float to the whole wavefront through SGPRs (k is wave-uniform): v_readlane instead of a
// ds_bpermute round trip, and the row address becomes scalar base + per-lane column offset (no 64-bit vector arithmetic)
__device__ __forceinline__ const float* bcast_ptr(const float* p, int k) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, k);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), k);
  return reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float bcast_f(float x, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k)); }
// W consecutive floats at p, no bounds: the caller clamps the column offset of lanes past the end of the row
template <int W>
__device__ __forceinline__ RowVec<W> row_load_raw(const float* __restrict__ p) {
  RowVec<W> r;
  if (W == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    r.a[0] = v.x; r.a[1] = v.y; r.a[2] = v.z; r.a[W - 1] = v.w;
  } else {
    const float2 v = *reinterpret_cast<const float2*>(p);
    r.a[0] = v.x; r.a[W - 1] = v.y;
  }
  return r;
}

// The kernel is bound by instruction issue, not by its gathers (ablated in the one-launch form, profiles/
// r03_attn_tile_phase_trace.txt: without the gathers its time does not change), so the per-key path is kept short:
//  * lane k resolves key k's three row addresses ONCE; per key they are broadcast through SGPRs (bcast_ptr) and every lane
//    adds its constant column offset - lanes past the end of a row read its first columns, which meet g = 0;
//  * node features are added as fma(fmask, yn, ya) with fmask = 0 when there is no table (yn then re-reads the node row),
//    edge features are emask * yb likewise;
//  * the range test of the time encoding (|x| <= 3e6: hardware cosine) is made once per key on a wave-uniform bound
//    instead of per element;
//  * the softmax exponentials are v_exp_f32 (arguments <= 0; ~2 ulp).
//  * FT = false: the model has neither a node-feature nor an edge-feature table (C4, C5): one gather per key instead of
//    three, and twice as many keys in flight.
// FS: feature streams gathered per key beside the node row - 2: node features + edge features, 1: edge features only
// (no node table, or the node part of a key comes from tg_model.c_table, which has the features folded in), 0: none
// diagnostic only (a build with -DTG_CORE_TRACE and TG_CORE_DBG=1, tools/trace_core.py - the stamps cost 29 registers, i.e. the
// third wavefront per SIMD, so they are compiled out of the production kernel): per-wavefront s_memtime stamps {entry, lists arrived, first key reduced,
// keys done, exit}.  C2 (one centre per wavefront, three wavefronts per SIMD): 6 450 ticks from entry to the lists' arrival,
// 5 600 to the first reduced key, 11 150 for the ten keys, 1 080 to the exit.  Hoisting the list / centre-id loads above the
// winners pass and the time-encoder rows and the G rows above the lists was tried: 178 registers (two wavefronts per SIMD),
// and held to 168 it spills and is slower (31 600 ticks against 24 300).
#ifdef TG_CORE_TRACE
__device__ unsigned long long g_core_trace[4096 * 5];
#define TG_CT(...) __VA_ARGS__
#else
#define TG_CT(...)
#endif
template <int NH, int NV, int W, int FS>
__global__ void __launch_bounds__(256) k_attn_core(tg_model m, int64_t Q, const float* __restrict__ ts,
                                                   const int64_t* __restrict__ l1_nids,
                                                   const int64_t* __restrict__ l1_eids, const float* __restrict__ l1_ts,
                                                   const float* __restrict__ reprs, const uint64_t* __restrict__ bm,
                                                   const uint32_t* __restrict__ rank, const float* __restrict__ G,
                                                   float* __restrict__ S, uint8_t* __restrict__ valid, DropCfg dc,
                                                   float* __restrict__ rsum, int direct, PosArgs pos,
                                                   const float* __restrict__ key_rows, const float* __restrict__ zl,
                                                   const float* __restrict__ gtab, const int64_t* __restrict__ cnids,
                                                   const float* __restrict__ ctab) {
  constexpr bool FT = FS > 0;
  using V = RowVec<W>;
  const int lane = lane_id();
  TG_CT(const bool trace = (direct & 256) != 0;)
  direct &= 1;
  TG_CT(const unsigned gw = blockIdx.x * 4 + (threadIdx.x >> 6);)
  TG_CT(unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0;)
  TG_CT(if (trace) tr0 = __builtin_amdgcn_s_memtime();)
  // direct (eager updates): neighbour rows come from the state tables, row(v) = has_msg[v] ? pending[v] : right[v];
  // the second dedup pass of the step rides here (a few thousand threads of work)
  if (pos.best) pos_winners_pass(pos, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
  if (pos.advance_off && blockIdx.x == 0 && threadIdx.x == 0) *pos.advance_off += pos.B;  // (the sampler has read it)
  const uint64_t dkey = drop_key(dc);
  const int d = m.d, de = m.d_e, K = m.n_neighbors;
  const int kvw = 2 * d + de;
  V w4[NV], p4[NV];
  int coff[NV], eoff[NV];  // column offsets of this lane in a node-width / edge-width row (0 past the end)
  float wmax = 0.f, pmax = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (lane + v * TG_WAVE) * W;
    coff[v] = c < d ? c : 0;
    eoff[v] = c < de ? c : 0;
    w4[v] = row_load<W>(m.te_freq, c, d, zl);
    p4[v] = row_load<W>(m.te_phase, c, d, zl);
#pragma unroll
    for (int j = 0; j < W; ++j) {
      wmax = fmaxf(wmax, fabsf(w4[v].a[j]));
      pmax = fmaxf(pmax, fabsf(p4[v].a[j]));
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    wmax = fmaxf(wmax, __shfl_xor(wmax, o, TG_WAVE));
    pmax = fmaxf(pmax, __shfl_xor(pmax, o, TG_WAVE));
  }
  const bool feat = FS == 2 && m.nfeats && !key_rows && !ctab;
  const float fmask = feat ? 1.f : 0.f;
  const float emask = m.efeats ? 1.f : 0.f;  // no edge table: the edge segment of a key row is zeros (feature_getter.py:95-99)
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < Q; i += (int64_t)gridDim.x * 4) {
    // ---- per-key metadata and row addresses, one key per lane
    int64_t nb_l = 0, eid_l = 0;
    float dt_l = 0.f;
    int u_l = 0;
    if (lane < K) {
      nb_l = l1_nids[i * K + lane];
      eid_l = l1_eids[i * K + lane];
      dt_l = ts[i] - l1_ts[i * K + lane];
      if (nb_l != 0) {
        if (ctab) {  // the node row comes from the per-node table: the has-message bit only feeds the invariant check
          if (pos.chk_err && bm_test(m.has_msg, nb_l)) check_msg_times(m, nb_l, pos.chk_err);
        } else if (direct) {
          const int64_t r = state_row(m, nb_l);
          u_l = (int)(2 * r + (bm_test(m.has_msg, r) ? 1 : 0));
          if (pos.chk_err && (u_l & 1)) check_msg_times(m, r, pos.chk_err);
        } else {
          u_l = (int)bm_rank(bm, rank, nb_l);
        }
      }
    }
    // key_rows (second attention layer of --n_layers 2): the node part of key k of centre i is row i*K + k of a dense
    // tensor - the neighbour's own embedding (temporal_agg_modules.py:57-66) - instead of its memory row + features.
    // A padding key (id 0) addresses row 0 of every table, which exists; its rows are fetched and never used.
    const int kk = lane < K ? lane : 0;
    const float* pn_l = key_rows ? key_rows + (i * K + kk) * d
                        : ctab   ? ctab + nb_l * d
                                 : (direct ? ((u_l & 1) ? m.pending_vals : m.right_vals) + (int64_t)(u_l >> 1) * d
                                           : reprs + (int64_t)u_l * d);
    const float* pf_l = feat ? m.nfeats + nb_l * d : pn_l;
    const float* pe_l = m.efeats ? m.efeats + eid_l * de : pn_l;
    unsigned long long live = __ballot(nb_l != 0);  // padding keys are masked (temporal_agg_modules.py:80)
    const bool any = live != 0ull;
    TG_CT(if (trace && tr1 == 0) { __builtin_amdgcn_sched_barrier(0); tr1 = __builtin_amdgcn_s_memtime() + (live & 0ull); })
    V g[NH][3][NV], acc[NH][3][NV];
    float mx[NH], l[NH], lk[NH];  // lk: sum of the kept exponentials (dropout), same rescaling as l
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      mx[h] = -INFINITY;
      l[h] = 0.f;
      lk[h] = 0.f;
      // eager query rows (tg_model.g_table): the centre NODE's row of the table instead of row i of this batch's product
      const float* gh = gtab ? gtab + (state_row(m, cnids[i]) * NH + h) * (int64_t)kvw : G + ((int64_t)i * NH + h) * kvw;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (lane + v * TG_WAVE) * W;
        g[h][0][v] = row_load<W>(gh, c, d, zl);  // zeros past the end of a segment: those lanes add nothing to a score
        g[h][1][v] = row_load<W>(gh + d, c, de, zl);
        g[h][2][v] = row_load<W>(gh + d + de, c, d, zl);
#pragma unroll
        for (int j = 0; j < W; ++j) acc[h][0][v].a[j] = acc[h][1][v].a[j] = acc[h][2][v].a[j] = 0.f;
      }
    }
    // Raw rows of the next keys travel in a ring of PD register slots while the current key is reduced.  Keys are
    // reduced in list order whatever PD is, so the result does not depend on it.
    // (measured: narrow rows, W = 2, gain from a fourth slot - C4 core 84 -> 78 us - and nothing from a fifth or sixth,
    // eight are slower; W = 4 is the same with three and four)
    constexpr int PD = FS == 2 ? (NV == 1 ? (W == 2 ? 4 : 3) : 2) : FS == 1 ? (NV == 1 ? 4 : 2) : (NV == 1 ? 6 : 3);
    constexpr int PF = FT ? PD : 1;       // slots of the edge-feature rows (none without tables)
    constexpr int PN = FS == 2 ? PD : 1;  // ... of the node-feature rows
    V ya[PD][NV], yn[PN][NV], yb[PF][NV];
    auto fetch = [&](int slot, int k) {
      const float* pn = bcast_ptr(pn_l, k);
#pragma unroll
      for (int v = 0; v < NV; ++v) ya[slot][v] = row_load_raw<W>(pn + coff[v]);
      if (FS == 2) {
        const float* pf = bcast_ptr(pf_l, k);
#pragma unroll
        for (int v = 0; v < NV; ++v) yn[slot][v] = row_load_raw<W>(pf + coff[v]);
      }
      if (FT) {
        const float* pe = bcast_ptr(pe_l, k);
#pragma unroll
        for (int v = 0; v < NV; ++v) yb[slot][v] = row_load_raw<W>(pe + eoff[v]);
      }
    };
    auto reduce = [&](int slot, int k) {
      const float dt = bcast_f(dt_l, k);
      // |dt w + phi| <= |dt| wmax + pmax: below the switch-over of time_enc_fast the hardware cosine serves every element
      const bool small = fmaf(fabsf(dt), wmax, pmax) < 2.9e6f;
      V x[3][NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (lane + v * TG_WAVE) * W;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          x[0][v].a[j] = FS == 2 ? fmaf(fmask, yn[slot][v].a[j], ya[slot][v].a[j]) : ya[slot][v].a[j];
          x[1][v].a[j] = FT ? emask * yb[slot][v].a[j] : 0.f;
        }
        if (small) {
#pragma unroll
          for (int j = 0; j < W; ++j) x[2][v].a[j] = cos_hw(__fadd_rn(__fmul_rn(dt, w4[v].a[j]), p4[v].a[j]));
        } else {
#pragma unroll
          for (int j = 0; j < W; ++j) x[2][v].a[j] = c + j < d ? time_enc_fast(dt, w4[v].a[j], p4[v].a[j]) : 0.f;
        }
      }
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        float p = 0.f;
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
          for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < W; ++j) p = fmaf(g[h][sgm][v].a[j], x[sgm][v].a[j], p);
        p = wave_sum(p);  // wave-uniform
        float b = 1.f;
        if (p > mx[h]) {  // new running maximum: rescale what has been accumulated (uniform branch)
          const float a = __expf(mx[h] - p);
          l[h] *= a;
          lk[h] *= a;
#pragma unroll
          for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
              for (int j = 0; j < W; ++j) acc[h][sgm][v].a[j] *= a;
          mx[h] = p;
        } else {
          b = __expf(p - mx[h]);
        }
        l[h] += b;
        if (dc.p > 0.f) {  // attention dropout (nn.MultiheadAttention): the softmax normaliser keeps every key
          b = drop_keep(dkey, DROP_ATTN, ((uint64_t)i * NH + h) * (uint64_t)K + (uint64_t)k, dc.thresh) ? b * dc.scale : 0.f;
          lk[h] += b;
        }
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
          for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < W; ++j) acc[h][sgm][v].a[j] = fmaf(b, x[sgm][v].a[j], acc[h][sgm][v].a[j]);
      }
    };
    // Keys are walked in list order, padding included, PD at a time; key k travels in ring slot k % PD.  Every fetch is
    // unconditional and only the arithmetic is skipped for padding keys: a branch that holds vector-memory instructions
    // makes the compiler's wait-count pass wait for EVERYTHING in flight at the join.
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) fetch(sl, min(sl, K - 1));
    for (int k0 = 0; k0 < K; k0 += PD) {
#pragma unroll
      for (int sl = 0; sl < PD; ++sl) {
        const int k = k0 + sl;
        if (k < K && ((live >> k) & 1ull)) reduce(sl, k);
        TG_CT(if (trace && k == 0 && tr2 == 0) { __builtin_amdgcn_sched_barrier(0); tr2 = __builtin_amdgcn_s_memtime() + (__float_as_uint(l[0]) & 0u); })
        fetch(sl, min(k + PD, K - 1));
      }
    }
    TG_CT(if (trace && tr3 == 0) { __builtin_amdgcn_sched_barrier(0); tr3 = __builtin_amdgcn_s_memtime() + (__float_as_uint(l[0]) & 0u); })
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float inv = any ? 1.f / l[h] : 0.f;
      float* sh = S + ((int64_t)i * NH + h) * kvw;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (lane + v * TG_WAVE) * W;
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
          for (int j = 0; j < W; ++j) acc[h][sgm][v].a[j] *= inv;
        row_store<W>(sh, c, d, acc[h][0][v]);
        row_store<W>(sh + d, c, de, acc[h][1][v]);
        row_store<W>(sh + d + de, c, d, acc[h][2][v]);
      }
    }
    if (lane == 0) {
      valid[i] = any ? 1 : 0;
      if (rsum) {
#pragma unroll
        for (int h = 0; h < NH; ++h) rsum[i * NH + h] = dc.p > 0.f ? (any ? lk[h] / l[h] : 0.f) : 1.f;
      }
    }
  }
#ifdef TG_CORE_TRACE
  if (trace && lane == 0 && gw < 4096) {
    g_core_trace[gw * 5 + 0] = tr0; g_core_trace[gw * 5 + 1] = tr1; g_core_trace[gw * 5 + 2] = tr2;
    g_core_trace[gw * 5 + 3] = tr3; g_core_trace[gw * 5 + 4] = __builtin_amdgcn_s_memtime();
  }
#endif
}
#ifdef TG_CORE_TRACE
extern "C" int tg_debug_core_trace(unsigned long long* out_host, int n_waves) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_core_trace), sizeof(unsigned long long) * 5 * n_waves) == hipSuccess ? 0 : -4;
}
#endif

int attn_dims_ok(const tg_model* m) {
  if (!m || m->d <= 0 || (m->d % 4) || m->d_e <= 0 || (m->d_e % 4) || m->n_neighbors <= 0) return 0;
  if (m->n_neighbors > TG_WAVE) return 0;  // one key per lane in k_attn_core
  if (m->n_head <= 0 || (2 * m->d) % m->n_head || ((2 * m->d / m->n_head) % 4)) return 0;
  return 1;
}

static int gru_split_knob() {
  static const int k = getenv("TG_GRU_SPLIT") ? atoi(getenv("TG_GRU_SPLIT")) : 0;  // tuning knob (0 = off, the default)
  return k;
}
static bool carve_attn(const tg_model* m, int64_t Q, Carver& cv, AttnWs& w) {
  const int d = m->d, kvw = 2 * m->d + m->d_e, nh = m->n_head;
  w.cc = cv.take<float>((size_t)Q * d);
  w.qp = cv.take<float>((size_t)Q * 2 * d);
  w.g = cv.take<float>((size_t)Q * nh * kvw);
  w.s = cv.take<float>((size_t)Q * nh * kvw);
  w.o = cv.take<float>((size_t)Q * 2 * d);
  w.hh = cv.take<float>((size_t)Q * 2 * d);
  w.t = cv.take<float>((size_t)Q * d);
  w.qconst = cv.take<float>((size_t)2 * d);
  w.rsum = cv.take<float>((size_t)Q * nh);
  w.valid = cv.take<uint8_t>((size_t)Q);
  w.sk = cv.take<float>(TG_SK_WS_FLOATS);
  return cv.ok;
}

static size_t attn_ws_bytes(const tg_model* m, int64_t Q) {
  const size_t d = m->d, kvw = 2 * m->d + m->d_e, nh = m->n_head;
  return align16(Q * d * 4) * 2 + align16(Q * 2 * d * 4) * 3 + align16(Q * nh * kvw * 4) * 2 + align16(2 * d * 4) +
         align16(Q * nh * 4) + align16(Q) + align16(TG_SK_WS_FLOATS * 4);
}

}  // namespace tg
static inline void prof_mark(tg_profiler* p, int i, hipStream_t st);
namespace tg {
constexpr int ST_ATTN_FIRST = 5;  // == ST_ATTN_PREP (checked by a static_assert below)

// ---- inference with pre-multiplied weights (tg_attn_fuse) --------------------------------
struct FusedView {
  const float *wqk, *gconst, *w1f, *b1, *c1;
  int nk;  // n_head * kvw
};
static FusedView fused_view(const tg_model* m, const float* f) {
  FusedView v{};
  const int d = m->d;
  v.nk = m->n_head * (2 * d + (m->efeats ? m->d_e : 0));  // compact form without an edge table (tg_fuse.hip)
  v.wqk = f;
  v.gconst = v.wqk + (size_t)v.nk * d;
  v.w1f = v.gconst + v.nk;
  v.b1 = v.w1f + (size_t)d * (v.nk + d);
  v.c1 = v.b1 + d;
  return v;
}

int attn_tile_launch(const tg_model* m, int64_t Q, const float* cc, const float* ts, const int64_t* l1_nids,
                     const int64_t* l1_eids, const float* l1_ts, const float* reprs, const uint64_t* bm,
                     const uint32_t* rank, float* out, int direct, const PosArgs* pos, hipStream_t st);

void launch_attn_core(const tg_model* m, int64_t Q, const float* ts, const int64_t* l1_nids, const int64_t* l1_eids,
                      const float* l1_ts, const float* reprs, const uint64_t* bm, const uint32_t* rank, const AttnWs& w,
                      const DropCfg& dc, hipStream_t st, int* rc_out, int direct = 0, const PosArgs* pos = nullptr,
                      const float* key_rows = nullptr, const float* gtab = nullptr, const int64_t* cnids = nullptr,
                      const float* ctab = nullptr) {
  const int d = m->d, d_e = m->d_e, nh = m->n_head;
  *rc_out = TG_OK;
  // Columns per lane: float4 (three columns per lane fill 58 of 64 lanes at d = 172 instead of 43 but measured SLOWER,
  // 29.7 vs 24.3 us at C2: 12-byte accesses straddle 16-byte sectors; that variant is gone).
  const int wmax = std::max(d, d_e);
  int W = 4, nv = (int)cdiv(cdiv(wmax, 4), TG_WAVE);
  {
    static const int w_knob = getenv("TG_ATTN_W") ? atoi(getenv("TG_ATTN_W")) : 0;  // tuning knob: 4 forces float4 lanes
    // narrow rows (d <= 128, e.g. LastFM's --dim 100): two columns per lane instead of four fill 50 lanes instead of 25;
    // the kernel is bound by per-key VALU work and latency there, not by bytes (8-byte accesses stay sector aligned)
    if (wmax <= 128 && w_knob != 4) { W = 2; nv = 1; }
  }
  const unsigned cgrid = flat_grid(Q, 4);
  static const int core_dbg = getenv("TG_CORE_DBG") ? atoi(getenv("TG_CORE_DBG")) : 0;  // diagnostic: s_memtime stamps (tools/trace_core.py)
  if (core_dbg) direct |= 256;
  const float* zl = zero_line();
  if (!zl) { *rc_out = TG_EHIP; return; }
  // feature streams per key: node + edge tables (2), the edge table alone - no node table, or the node rows come from the
  // per-node table of centre rows with the features folded in (1) - or none (0)
  const int fs = (m->nfeats && !key_rows && !ctab) ? 2 : (m->efeats ? 1 : 0);
#define TG_CORE_FS(NH_, NV_, W_, FS_)                                                                                      \
  TG_KLAUNCH((k_attn_core<NH_, NV_, W_, FS_>), dim3(cgrid), dim3(256), 0, st, *m, Q, ts, l1_nids, l1_eids, l1_ts,          \
                     reprs, bm, rank, (const float*)w.g, w.s, w.valid, dc, dc.p > 0.f ? w.rsum : (float*)nullptr, direct,  \
                     pos ? *pos : PosArgs{}, key_rows, zl, gtab, cnids, ctab)
#define TG_CORE(NH_, NV_, W_)                  \
  do {                                         \
    if (fs == 2) TG_CORE_FS(NH_, NV_, W_, 2);  \
    else if (fs == 1) TG_CORE_FS(NH_, NV_, W_, 1); \
    else TG_CORE_FS(NH_, NV_, W_, 0);          \
  } while (0)
  if (nh == 2 && nv == 1 && W == 2) TG_CORE(2, 1, 2);
  else if (nh == 1 && nv == 1 && W == 2) TG_CORE(1, 1, 2);
  else if (nh == 4 && nv == 1 && W == 2) TG_CORE(4, 1, 2);
  else if (nh == 2 && nv == 1) TG_CORE(2, 1, 4);
  else if (nh == 2 && nv == 2) TG_CORE(2, 2, 4);
  else if (nh == 1 && nv == 1) TG_CORE(1, 1, 4);
  else if (nh == 4 && nv == 1) TG_CORE(4, 1, 4);
  else *rc_out = TG_EUNSUPPORTED;
#undef TG_CORE_FS
#undef TG_CORE
}

// ---- side lane: a second HIP stream per device for work that is independent of the launches beside it -----------------
// At large batches the step's launches are long and bound by different resources: the products and the updater by the
// matrix pipe, the write-back and the sampler by memory round trips.  Riders (workgroups of the same launch) were measured
// to cost such products more than they save (see gemm_launch); a forked stream lets the dispatcher co-schedule the two
// kernels' workgroups instead.  fork: the lane waits for everything enqueued on `st` so far; join: `st` waits for the lane.
// Under stream capture (a hipGraph of several steps) the event pair makes the lane part of the capture: a parallel branch
// of the graph.  The lane is created by the first call outside a capture (every caller runs a step eagerly first).
// MEASURED at C5 shape (B = 65 536, d = 256, 1x MI355X, 30 steps): NOT faster - 3.905 ms per step against 3.861 ms with
// everything on one stream: fc1 grows by the write-back's own duration (1.10 -> 1.28 ms), the updater and the query rows
// by the sampler's (1.13 -> 1.23, 0.40 -> 0.54 ms incl. the centres launch).  The matrix kernels fill the register file
// (k_gru<4, 1>: 2 x 256 registers per SIMD lane, k_gemm_rb<2, 2>: 2 x 228), so the side kernel's wavefronts only get
// slots the matrix kernel's blocks give up - time slicing, not overlap.  Hence OFF by default; TG_SIDE_STREAM=1 switches
// the form on (read at every call: tests/test_hip_timed_form.py runs it against the oracle).
struct SideLane {
  hipStream_t s = nullptr;
  hipEvent_t fork[2] = {nullptr, nullptr}, join[2] = {nullptr, nullptr};
  bool ok = false;
};
static SideLane* side_lane(hipStream_t st) {
  const char* knob = getenv("TG_SIDE_STREAM");  // tuning knob (read per call; default off, see above)
  if (!knob || atoi(knob) == 0) return nullptr;
  static SideLane lanes[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideLane& L = lanes[dev];
  if (L.ok) return &L;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;  // not now: no stream / event is created inside a capture
  }
  bool good = hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 2 && good; ++i)
    good = hipEventCreateWithFlags(&L.fork[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&L.join[i], hipEventDisableTiming) == hipSuccess;
  if (!good) {
    (void)hipGetLastError();
    return nullptr;
  }
  L.ok = true;
  return &L;
}
static bool lane_fork(SideLane* L, int i, hipStream_t st) {
  return hipEventRecord(L->fork[i], st) == hipSuccess && hipStreamWaitEvent(L->s, L->fork[i], 0) == hipSuccess;
}
static bool lane_join(SideLane* L, int i, hipStream_t st) {
  return hipEventRecord(L->join[i], L->s) == hipSuccess && hipStreamWaitEvent(st, L->join[i], 0) == hipSuccess;
}
// the write-back rider's pass (STEP 4-5 and the bookkeeping of STEP 6: writeback_fused_body<false>) as a launch of its own
__global__ void __launch_bounds__(256) k_wb_rider(WbRider r) { r.run(blockIdx.x); }
constexpr int64_t SIDE_MIN_B = 16384;  // batches above this: launches that host no riders (gemm_launch, step_forward)

static void launch_centres(const tg_model* m, int64_t Q, const int64_t* nids, const float* reprs, const uint64_t* bm,
                           const uint32_t* rank, const AttnWs& w, const PosArgs* pos, const DirectArgs* da, hipStream_t st,
                           bool no_copy = false) {
  const int d = m->d;
  if (da)  // rows from the state tables; checks + first dedup pass ride along (the second one rides on the core)
    // (no_copy: the centre rows are read from the per-node table, tg_model.c_table - only checks, dedup and snapshot)
    hipLaunchKernelGGL(k_attn_centres_direct, dim3(flat_grid((no_copy ? std::max<int64_t>(da->n_snap, 1) : Q) * (d / 4), 256)),
                       dim3(256), 0, st, *m, Q, nids, (const float4*)m->nfeats, no_copy ? (float4*)nullptr : (float4*)w.cc, *da,
                       pos ? *pos : PosArgs{});
  else
    hipLaunchKernelGGL(k_attn_centres, dim3(flat_grid(Q * (d / 4), 256)), dim3(256), 0, st, Q, d / 4, nids,
                       (const float4*)reprs, bm, rank, (const float4*)m->nfeats, (float4*)w.cc, 0, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
                       pos ? *pos : PosArgs{});
}

static int attn_forward_fused(const tg_model* m, int64_t Q, const int64_t* nids, const float* ts,
                              const int64_t* l1_nids, const int64_t* l1_eids, const float* l1_ts, const float* reprs,
                              const uint64_t* bm, const uint32_t* rank, float* out, const AttnWs& w, hipStream_t st,
                              tg_profiler* pf, const PosArgs* pos, const DirectArgs* da, bool centres_done,
                              const float* key_rows, bool use_gtab, const WbRider* wbr, bool* wb_rode, GruSplit* gs,
                              const CollateRider* sampler, bool* sampler_rode) {
  // stage numbering of the profiler is kept: q -> "merged q+g", g -> skipped, v/out -> skipped, fc1 -> fused
  int stage = ST_ATTN_FIRST + 1;
  const int d = m->d;
  const FusedView f = fused_view(m, m->attn_fused);
  if (!centres_done)  // else: rode on the sampler's launch (or on the previous step's last one)
    launch_centres(m, Q, nids, reprs, bm, rank, w, pos, da, st, use_gtab && m->c_table && da);
  int rc;
  if (attn_tile_applies(m) && !key_rows && !use_gtab) {  // the whole block in one launch, G and S in LDS only (tg_attn_tile.hip); timed as the core
    prof_mark(pf, stage++, st);
    prof_mark(pf, stage++, st);
    prof_mark(pf, stage++, st);
    if ((rc = attn_tile_launch(m, Q, w.cc, ts, l1_nids, l1_eids, l1_ts, reprs, bm, rank, out, da ? 1 : 0,
                               da ? pos : nullptr, st)) != TG_OK)
      return rc;
    for (int i = 0; i < 5; ++i) prof_mark(pf, stage++, st);
    return check_launch("tg_temporal_attn_fwd(tile)");
  }
  GemmArgs g{};
  // G = c Wqk^T + gconst   (scaled query folded through the key projection, all heads at once) - or, with eager query
  // rows, nothing: the core reads G of a centre from the per-node table
  prof_mark(pf, stage++, st);
  if (!use_gtab) {
    g.m_cap = Q; g.n = f.nk; g.k = d; g.a0 = ASeg{w.cc, d, d, nullptr};
    g.w = f.wqk; g.ldw = d; g.bias = f.gconst; g.c = w.g; g.ldc = f.nk; g.alpha = 1.f; g.nbatch = 1;
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  }
  prof_mark(pf, stage++, st);
  prof_mark(pf, stage++, st);
  tg_model mc = *m;  // without an edge table the fused weights are compact: the key rows have no edge segment
  if (!m->efeats) mc.d_e = 0;
  {
    KSlot ks_(KT_CORE);
    launch_attn_core(&mc, Q, ts, l1_nids, l1_eids, l1_ts, reprs, bm, rank, w, DropCfg{}, st, &rc, da ? 1 : 0, da ? pos : nullptr,
                     key_rows, use_gtab ? m->g_table : nullptr, nids, (use_gtab && da) ? m->c_table : nullptr);
  }
  if (rc != TG_OK) return rc;
  prof_mark(pf, stage++, st);
  prof_mark(pf, stage++, st);
  // t = relu([S | c] W1f^T + b1 + valid * c1)   (value projection, out projection and fc1 merged)
  prof_mark(pf, stage++, st);
  g = GemmArgs{};
  g.m_cap = Q; g.n = d; g.k = f.nk + d;
  g.a0 = ASeg{w.s, f.nk, f.nk, nullptr};
  g.a1 = (use_gtab && m->c_table) ? ASeg{m->c_table, d, d, nids} : ASeg{w.cc, d, d, nullptr};  // centre rows: table or copy
  g.w = f.w1f; g.ldw = f.nk + d; g.bias = f.b1; g.bias2 = f.c1; g.bias2_valid = w.valid;
  g.c = w.t; g.ldc = d; g.relu = 1; g.alpha = 1.f; g.nbatch = 1;
  // C2-sized batches leave this product with fewer tiles than CUs: dealt as stream-K pieces, which fc2 sums
  // (+ b1 + valid * c1, ReLU) while it stages its A operand; otherwise the plain product writes t
  // ... or, with few enough 48 x 48 tiles, LDS-free K-split blocks that leave t itself (k_gemm_ks16): fc2 is then a plain
  // short-K product
  SkPlan sk{};
  KSlot ks_fc1(KT_FC1);
  {
    // fc1 and fc2 as ONE launch where the batch makes 128 .. 232 blocks of 16 whole rows (C2: 192; tg_gemm.hip:
    // k_gemm_ks16_fc2): fc2 runs as the epilogue of the block that owns the rows, STEP 6's rows leave it, the write-back
    // rider takes the CUs the product leaves idle
    const bool ext12 = wbr && wbr->planned0;
    const bool own12 = !ext12 && wbr && pos && pos->win_row;
    GemmArgs g2{};
    g2.m_cap = Q; g2.n = d; g2.k = d;
    g2.a0 = ASeg{w.t, d, d, nullptr};
    g2.w = m->attn_fc2.w; g2.ldw = d; g2.bias = m->attn_fc2.b;
    g2.c = out; g2.ldc = d; g2.alpha = 1.f; g2.nbatch = 1;
    if (own12) { g2.c2 = m->left_vals; g2.c2_rows = pos->win_row; g2.c2_m = 2 * wbr->a.B; g2.ldc2 = d; }
    bool rode12 = false;
    if (!gs && !sampler && gemm_fc12_launch(g, g2, st, (ext12 || own12) ? wbr : nullptr, &rode12)) {
      prof_mark(pf, stage++, st);
      if (wb_rode) *wb_rode = rode12;
      return check_launch("tg_temporal_attn_fwd(fused, fc1 + fc2)");
    }
  }
  bool wb_on_fc1 = false;  // the write-back rider on the fc1 launch: then fc2 only stores STEP 6's rows (c2)
  bool gi_rode = false;  // the split updater's input-side product as a second problem of this launch (variant 1)
  const bool ext = wbr && wbr->planned0;  // a caller's rider (tg_part_step): hosted like the write-back rider, no second row copy
  const bool ks16 = gemm_ks16_launch(g, st, (ext || (wbr && pos && pos->win_row)) ? wbr : nullptr, &wb_on_fc1,
                                     (gs && gs->variant == 1) ? &gs->gi : nullptr, &gi_rode);
  const bool pieces = !ks16 && gemm_sk_partials(g, w.sk, TG_SK_WS_FLOATS, st, &sk);
  // large batch (this product hosts no rider): the write-back rider as a launch of its own on the side lane, beside this
  // product - STEP 4-5 read the snapshot and the winners the core's launch left, nothing this product touches - joined in
  // front of fc2, whose epilogue stores STEP 6's rows
  SideLane* lane = nullptr;
  bool wb_side = false;
  if (!ks16 && !pieces && !ext && wbr && pos && pos->win_row && !gs && wbr->a.B > SIDE_MIN_B && (lane = side_lane(st)) != nullptr) {
    if (!lane_fork(lane, 0, st)) return TG_EHIP;
    WbRider wr = *wbr;
    wr.blocks = flat_grid(2 * wr.a.B, 4);
    wr.last = 0u;
    hipLaunchKernelGGL(k_wb_rider, dim3(wr.blocks), dim3(256), 0, lane->s, wr);
    wb_side = true;
  }
  if (!ks16 && !pieces && (rc = gemm_launch(g, st)) != TG_OK) return rc;
  if (wb_side && !lane_join(lane, 0, st)) return TG_EHIP;
  prof_mark(pf, stage++, st);
  KSlot ks_fc2(KT_FC2);
  g = GemmArgs{};
  g.m_cap = Q; g.n = d; g.k = d;
  g.a0 = ASeg{w.t, d, d, nullptr};
  if (pieces) {
    g.ask_part = sk.part; g.ask_U = sk.U; g.ask_nkt = sk.nkt; g.ask_NT = sk.NT; g.ask_pieces = sk.pieces;
    g.ask_bias = f.b1; g.ask_bias2 = f.c1; g.ask_valid = w.valid; g.ask_relu = 1; g.ask_alpha = 1.f;
  }
  g.w = m->attn_fc2.w; g.ldw = d; g.bias = m->attn_fc2.b;
  g.c = out; g.ldc = d; g.alpha = 1.f; g.nbatch = 1;
  bool rode = false;
  if (ext) {
    // (no second destination; hosted by fc1 already, or by this launch, or not at all)
  } else if (wbr && pos && pos->win_row) {  // STEP 6's rows leave this product's epilogue; STEP 4-5 ride on its launch (WbRider)
    g.c2 = m->left_vals; g.c2_rows = pos->win_row; g.c2_m = 2 * wbr->a.B; g.ldc2 = d;
  } else {
    wbr = nullptr;
  }
  if (wb_on_fc1 && gs && gs->variant == 2) {
    // ... and the split updater's W_ih msg shares THIS launch (variant 2): fc2's 144 blocks leave 112 CUs idle at C2 and its
    // other blocks finish early; the tail then is a short launch of its own behind it (step_writeback_b)
    bool gi_here = false;
    if ((rc = gemm_launch(g, st, nullptr, &gi_here, nullptr, nullptr, &gs->gi)) != TG_OK) return rc;
    gs->gi_done = gi_here;
    rode = true;
  } else if (wb_on_fc1 && gi_rode) {  // ... and the updater's tail shares this launch (it reads t, like fc2)
    bool tail_rode = false;
    if ((rc = gemm_launch(g, st, nullptr, &tail_rode, nullptr, &gs->tail)) != TG_OK) return rc;
    if (!tail_rode && (rc = gru_tail_launch(gs->tail, st)) != TG_OK) return rc;
    gs->done = true;
    rode = true;
  } else if (wb_side) {  // STEP 4-5 ran beside fc1 (side lane): this product only stores STEP 6's rows (c2)
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
    rode = true;
  } else if (wb_on_fc1) {  // ... or rode on fc1's already: this launch is free to host the NEXT batch's sampler (collate
    // prefetch; the stream offset has been advanced on fc1's launch, nothing from here on reads this batch's query arrays
    // or neighbour lists)
    bool srode = false;
    if (sampler && !ext) {
      CollateRider cs = *sampler;
      cs.parts = 1u;
      if ((rc = gemm_launch(g, st, nullptr, &srode, &cs)) != TG_OK) return rc;
    } else if ((rc = gemm_launch(g, st)) != TG_OK) {
      return rc;
    }
    if (sampler_rode) *sampler_rode = srode;
    rode = true;
  } else if ((rc = gemm_launch(g, st, wbr, &rode)) != TG_OK) {
    return rc;
  }
  if (wb_rode) *wb_rode = rode;
  return check_launch("tg_temporal_attn_fwd(fused)");
}

int attn_forward(const tg_model* m, int64_t Q, const int64_t* nids, const float* ts, const int64_t* l1_nids,
                 const int64_t* l1_eids, const float* l1_ts, const float* reprs, const uint64_t* bm,
                 const uint32_t* rank, float* out, const AttnWs& w, hipStream_t st, tg_profiler* pf = nullptr,
                 const DropCfg* drop = nullptr, const PosArgs* pos = nullptr, const DirectArgs* da = nullptr,
                 const float* key_rows = nullptr, bool centres_done = false, bool use_gtab = false,
                 const WbRider* wbr = nullptr, bool* wb_rode = nullptr, GruSplit* gs = nullptr,
                 const CollateRider* sampler = nullptr, bool* sampler_rode = nullptr) {
  if (wb_rode) *wb_rode = false;
  if (sampler_rode) *sampler_rode = false;
  if (gs) gs->done = false;
  const DropCfg dc = drop ? *drop : DropCfg{};
  int stage = ST_ATTN_FIRST;
  prof_mark(pf, stage++, st);
  const int d = m->d, d_e = m->d_e, kvw = 2 * d + d_e, nh = m->n_head, dh = 2 * d / nh, E = 2 * d;
  if (m->attn_fused && dc.p == 0.f)  // (the pre-multiplied weights do not care where the node part of a key row comes from)
    return attn_forward_fused(m, Q, nids, ts, l1_nids, l1_eids, l1_ts, reprs, bm, rank, out, w, st, pf, pos, da, centres_done,
                              key_rows, use_gtab && m->g_table && !key_rows, wbr, wb_rode, gs, sampler, sampler_rode);
  const int qblocks = (int)cdiv(2 * d, 4);
  if (da) {  // the constant half of the query projection from the rank-form kernel (no centre rows), then the direct centres
    hipLaunchKernelGGL(k_attn_centres, dim3(1 + qblocks), dim3(256), 0, st, (int64_t)0, d / 4, nids, (const float4*)reprs, bm,
                       rank, (const float4*)m->nfeats, (float4*)w.cc, qblocks, m->attn_wq, m->attn_b_in, m->te_freq,
                       m->te_phase, w.qconst, PosArgs{});
    if (!centres_done) launch_centres(m, Q, nids, reprs, bm, rank, w, pos, da, st);
  } else {
    hipLaunchKernelGGL(k_attn_centres, dim3(flat_grid(Q * (d / 4), 256) + qblocks), dim3(256), 0, st, Q, d / 4, nids,
                       (const float4*)reprs, bm, rank, (const float4*)m->nfeats, (float4*)w.cc, qblocks, m->attn_wq,
                       m->attn_b_in, m->te_freq, m->te_phase, w.qconst, pos ? *pos : PosArgs{});
  }
  int rc;
  GemmArgs g{};
  // q = (Wq [c | TE(0)] + bq) / sqrt(dh)          (F.multi_head_attention_forward scaling)
  prof_mark(pf, stage++, st);
  g = GemmArgs{};
  g.m_cap = Q; g.n = E; g.k = d;
  g.a0 = ASeg{w.cc, d, d, nullptr};
  g.w = m->attn_wq; g.ldw = E; g.bias = w.qconst;
  g.c = w.qp; g.ldc = E; g.alpha = 1.0f / sqrtf((float)dh); g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // g_h = Wk_h^T q_h   (k-major weight view: B[k][n] = Wk[h*dh + k][n])
  prof_mark(pf, stage++, st);
  g = GemmArgs{};
  g.m_cap = Q; g.n = kvw; g.k = dh;
  g.a0 = ASeg{w.qp, E, dh, nullptr}; g.a0_bs = dh;
  g.w = m->attn_wk; g.ldw = kvw; g.w_kmajor = 1; g.w_bs = (int64_t)dh * kvw;
  g.c = w.g; g.ldc = (int64_t)nh * kvw; g.c_bs = kvw; g.alpha = 1.f; g.nbatch = nh;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // gather + scores + softmax + weighted raw sum
  prof_mark(pf, stage++, st);
  {
    KSlot ks_(KT_CORE);
    launch_attn_core(m, Q, ts, l1_nids, l1_eids, l1_ts, reprs, bm, rank, w, dc, st, &rc, da ? 1 : 0, da ? pos : nullptr, key_rows);
  }
  if (rc != TG_OK) return rc;
  // o_h = Wv_h s_h + bv_h
  prof_mark(pf, stage++, st);
  g = GemmArgs{};
  g.m_cap = Q; g.n = dh; g.k = kvw;
  g.a0 = ASeg{w.s, (int64_t)nh * kvw, kvw, nullptr}; g.a0_bs = kvw;
  g.w = m->attn_wv; g.ldw = kvw; g.w_bs = (int64_t)dh * kvw;
  g.bias = m->attn_b_in + 2 * E; g.bias_bs = dh;
  if (dc.p > 0.f) { g.bias_rs = w.rsum; g.ld_brs = nh; }
  g.c = w.o; g.ldc = E; g.c_bs = dh; g.alpha = 1.f; g.nbatch = nh;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // h = Wo o + bo, zeroed for centres without neighbours (temporal_agg_modules.py:224-231)
  prof_mark(pf, stage++, st);
  g = GemmArgs{};
  g.m_cap = Q; g.n = E; g.k = E;
  g.a0 = ASeg{w.o, E, E, nullptr};
  g.w = m->attn_out.w; g.ldw = E; g.bias = m->attn_out.b;
  g.c = w.hh; g.ldc = E; g.row_valid = w.valid; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // z = fc2(relu(fc1([h | c])))   (MergeLayer, basic_modules.py:16-19)
  prof_mark(pf, stage++, st);
  g = GemmArgs{};
  g.m_cap = Q; g.n = d; g.k = E + d;
  g.a0 = ASeg{w.hh, E, E, nullptr}; g.a1 = ASeg{w.cc, d, d, nullptr};
  g.w = m->attn_fc1.w; g.ldw = E + d; g.bias = m->attn_fc1.b;
  g.c = w.t; g.ldc = d; g.relu = 1; g.alpha = 1.f; g.nbatch = 1;
  {
    KSlot ks_(KT_FC1);
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  }
  prof_mark(pf, stage++, st);
  KSlot ks_fc2(KT_FC2);
  g = GemmArgs{};
  g.m_cap = Q; g.n = d; g.k = d;
  g.a0 = ASeg{w.t, d, d, nullptr};
  g.w = m->attn_fc2.w; g.ldw = d; g.bias = m->attn_fc2.b;
  g.c = out; g.ldc = d; g.alpha = 1.f; g.nbatch = 1;
  bool rode = false;
  if (wbr && pos && pos->win_row && dc.p == 0.f) {  // the write-back rider (see attn_forward_fused)
    g.c2 = m->left_vals; g.c2_rows = pos->win_row; g.c2_m = 2 * wbr->a.B; g.ldc2 = d;
  } else {
    wbr = nullptr;
  }
  if ((rc = gemm_launch(g, st, wbr, &rode)) != TG_OK) return rc;
  if (wb_rode) *wb_rode = rode;
  return check_launch("tg_temporal_attn_fwd");
}

// ---- message transform + updater ------------------------------------------------------
struct ApplyWs {
  float *t0, *t1, *t2;
};

static size_t apply_ws_bytes(const tg_model* m, int64_t cap) {
  const size_t mw = 3 * (size_t)m->d + m->d_e;
  size_t b = 0;
  if (m->tsfm == TG_TSFM_MLP) b += align16(cap * (mw / 2) * 4);
  if (m->tsfm != TG_TSFM_ID) b += align16(cap * mw * 4);
  if (m->upd_fn == TG_UPD_MERGE) b += align16(cap * (size_t)m->d * 4);
  return b + 16;
}

int apply_messages(const tg_model* m, const int64_t* outdated, const int32_t* out_pos, const int32_t* n_dev,
                   int64_t cap, float* reprs, uint32_t* err, void* ws, size_t ws_bytes, hipStream_t st,
                   bool checked_already = false, float* gates = nullptr, int64_t rows_bound = 0, float* out2 = nullptr,
                   const float* add2 = nullptr, bool out2_by_row = false) {
  const int d = m->d, mw = 3 * m->d + m->d_e;
  Carver cv(ws, ws_bytes);
  ApplyWs w{};
  if (m->tsfm == TG_TSFM_MLP) w.t0 = cv.take<float>((size_t)cap * (mw / 2));
  if (m->tsfm != TG_TSFM_ID) w.t1 = cv.take<float>((size_t)cap * mw);
  if (m->upd_fn == TG_UPD_MERGE) w.t2 = cv.take<float>((size_t)cap * d);
  if (!cv.ok) return TG_EWORKSPACE;
  if (!checked_already)
    hipLaunchKernelGGL(k_check_messages, dim3(flat_grid(cap, 256)), dim3(256), 0, st, *m, outdated, n_dev, cap, err);
  int rc;
  ASeg x{m->msg_vals, mw, mw, outdated};  // raw messages gathered from the mailbox
  if (m->tsfm == TG_TSFM_LINEAR || m->tsfm == TG_TSFM_MLP) {
    if (m->tsfm == TG_TSFM_MLP && ((mw / 2) % 4)) return TG_EUNSUPPORTED;
    GemmArgs g{};
    g.m_cap = cap; g.m_dev = n_dev; g.k = mw; g.a0 = x; g.alpha = 1.f; g.nbatch = 1;
    g.w = m->tsfm1.w; g.ldw = mw; g.bias = m->tsfm1.b;
    if (m->tsfm == TG_TSFM_MLP) {
      g.n = mw / 2; g.c = w.t0; g.ldc = mw / 2; g.relu = 1;
      if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
      g = GemmArgs{};
      g.m_cap = cap; g.m_dev = n_dev; g.k = mw / 2; g.a0 = ASeg{w.t0, mw / 2, mw / 2, nullptr};
      g.w = m->tsfm2.w; g.ldw = mw / 2; g.bias = m->tsfm2.b; g.alpha = 1.f; g.nbatch = 1;
    }
    g.n = mw; g.c = w.t1; g.ldc = mw; g.relu = 0;
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
    x = ASeg{w.t1, mw, mw, nullptr};
  }
  const float* upd_vals = (m->upd_src == TG_SRC_LEFT) ? m->left_vals : m->right_vals;
  ASeg h{upd_vals, d, d, outdated};
  if (m->upd_fn == TG_UPD_GRU) {
    GruArgs a{};
    a.cap = cap; a.n_dev = n_dev; a.d = d; a.xw = mw; a.x = x; a.h = h;
    a.w_ih = m->gru_w_ih; a.w_hh = m->gru_w_hh; a.b_ih = m->gru_b_ih; a.b_hh = m->gru_b_hh;
    a.out = reprs; a.ldo = d; a.out_rows = out_pos; a.gates = gates; a.out2 = out2; a.add2 = add2;
    a.out2_by_row = out2_by_row ? 1 : 0;
    if (!m->efeats && m->tsfm == TG_TSFM_ID) {
      // raw mailbox rows [own | other | edge | time] without an edge table: the edge segment [2d, 2d + d_e) is zeros
      // (memory.py:91 over feature_getter.py:95-99); the k-tiles that lie entirely inside it are skipped
      const int first = (2 * d + 31) / 32, last = (2 * d + m->d_e) / 32;  // tiles [first, last) are inside
      if (last > first) { a.x_skip_at = first; a.x_skip_n = last - first; }
    }
    a.rows_hint = std::min<int64_t>(cap, m->n_nodes);
    if (rows_bound > 0) a.rows_hint = std::min<int64_t>(a.rows_hint, rows_bound);  // the caller's bound on the live rows
    return gru_launch(a, st);
  }
  GemmArgs g{};  // MergeUpdater: fc2(relu(fc1([msg | mem])))
  g.m_cap = cap; g.m_dev = n_dev; g.n = d; g.k = mw + d; g.a0 = x; g.a1 = h;
  g.w = m->upd_fc1.w; g.ldw = mw + d; g.bias = m->upd_fc1.b; g.c = w.t2; g.ldc = d; g.relu = 1; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  g = GemmArgs{};
  g.m_cap = cap; g.m_dev = n_dev; g.n = d; g.k = d; g.a0 = ASeg{w.t2, d, d, nullptr};
  g.w = m->upd_fc2.w; g.ldw = d; g.bias = m->upd_fc2.b; g.c = reprs; g.ldc = d; g.c_rows = out_pos; g.alpha = 1.f; g.nbatch = 1;
  return gemm_launch(g, st);
}

int apply_messages_rows(const tg_model* m, const int64_t* rows, const int32_t* rows32, const int32_t* n_dev, int64_t cap,
                        uint32_t* err, void* ws, size_t ws_bytes, hipStream_t st) {
  // (checked_already: the rows have just been written by STEP 5 / 6 of this very batch - the message / memory time
  //  invariants hold by construction, as for the single-GPU step's eager updater)
  return apply_messages(m, rows, rows32, n_dev, cap, m->pending_vals, err, ws, ws_bytes, st, true);
}

// unified positive-node dedup of the fused step: float32 timestamps, winner = latest ts,
// first position among ties.  best[rank(node)] = max over positions of (ts_key << 32 | ~pos).
__global__ void k_pos_max(int64_t B, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                          const float* __restrict__ ts, const uint64_t* __restrict__ bm,
                          const uint32_t* __restrict__ rank, unsigned long long* __restrict__ best,
                          int32_t* __restrict__ winner_count) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *winner_count = 0;  // k_pos_winners (next launch) counts into it
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * B; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i < B ? i : i - B;
    const int64_t node = i < B ? src[e] : dst[e];
    const unsigned long long key = (orderable(ts[e]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
    atomicMax(best + bm_rank(bm, rank, node), key);
  }
}

__global__ void k_pos_winners(int64_t B, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                              const float* __restrict__ ts, const uint64_t* __restrict__ bm,
                              const uint32_t* __restrict__ rank, const unsigned long long* __restrict__ best,
                              int64_t* __restrict__ upos, int64_t* __restrict__ index, int32_t* __restrict__ count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * B; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i < B ? i : i - B;
    const int64_t node = i < B ? src[e] : dst[e];
    const unsigned long long key = (orderable(ts[e]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
    if (best[bm_rank(bm, rank, node)] == key) {
      const int slot = atomicAdd(count, 1);
      upos[slot] = node;
      index[slot] = i;
    }
  }
}

// batch slice -> query arrays: nids3 = cat[src, dst, neg], ts3 = tile(ts, 3) (float64 for the
// sampler, float32 for the model, data_loader.py:79-81,92), eids copy.  `off` (nullable)
// is the device-resident element offset of the batch inside the stream arrays.
__global__ void k_build_queries(int64_t B, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                const int64_t* __restrict__ neg, const double* __restrict__ ts,
                                const int64_t* __restrict__ eids, const int64_t* __restrict__ off,
                                int64_t* __restrict__ nids3, double* __restrict__ ts3, float* __restrict__ ts3f,
                                int64_t* __restrict__ eids_b, uint32_t* __restrict__ tmin_key) {
  const int64_t o = off ? *off : 0;
  float tmin = INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * B; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i % B;
    const int r = (int)(i / B);
    nids3[i] = r == 0 ? src[o + e] : (r == 1 ? dst[o + e] : neg[o + e]);
    const double t = ts[o + e];
    ts3[i] = t;
    ts3f[i] = (float)t;
    if (r == 0) {
      eids_b[e] = eids[o + e];
      tmin = fminf(tmin, (float)t);
    }
  }
  if (tmin_key) {  // lazy restart only: the batch's earliest (float32) time, as sample_batch_body leaves it (tg_sample.h)
    __shared__ float s_tmin[4];
    for (int sh = 32; sh > 0; sh >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, sh, TG_WAVE));
    if (lane_id() == 0) s_tmin[threadIdx.x >> 6] = tmin;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float v = fminf(fminf(s_tmin[0], s_tmin[1]), fminf(s_tmin[2], s_tmin[3]));
      if (v < INFINITY) atomicMax(tmin_key, ~(uint32_t)orderable(v));
    }
  }
}

__global__ void k_advance(int64_t* off, int64_t B) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *off += B;
}

int unique_compact_launch(const uint8_t* flags, uint64_t* bm, int64_t n_nodes, uint32_t* rank, int64_t* ids,
                          int32_t* count, int64_t cap, const uint64_t* hm, uint32_t* rank2, int64_t* ids2,
                          int32_t* pos2, int32_t* count2, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace tg

using namespace tg;

extern "C" size_t tg_temporal_attn_workspace_bytes(const tg_model* m, int64_t Q) {
  if (!attn_dims_ok(m) || Q < 0) return 0;
  return attn_ws_bytes(m, Q) + 64;
}

extern "C" int tg_temporal_attn_fwd(const tg_model* m, int64_t Q, const int64_t* nids, const float* ts,
                                    const int64_t* l1_nids, const int64_t* l1_eids, const float* l1_ts,
                                    const float* reprs, const uint64_t* bitmap, const uint32_t* rank, float* out,
                                    void* ws, size_t ws_bytes, void* stream) {
  if (!attn_dims_ok(m) || Q < 0) return TG_EINVAL;
  if (Q == 0) return TG_OK;
  if (!nids || !ts || !l1_nids || !l1_eids || !l1_ts || !reprs || !bitmap || !rank || !out) return TG_EINVAL;
  Carver cv(ws, ws_bytes);
  AttnWs w{};
  if (!carve_attn(m, Q, cv, w)) return TG_EWORKSPACE;
  return attn_forward(m, Q, nids, ts, l1_nids, l1_eids, l1_ts, reprs, bitmap, rank, out, w, as_stream(stream));
}

extern "C" int tg_temporal_attn_fwd_keys(const tg_model* m, int64_t Q, const int64_t* nids, const float* ts,
                                         const int64_t* l1_nids, const int64_t* l1_eids, const float* l1_ts,
                                         const float* reprs, const uint64_t* bitmap, const uint32_t* rank,
                                         const float* key_rows, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!attn_dims_ok(m) || Q < 0) return TG_EINVAL;
  if (Q == 0) return TG_OK;
  if (!nids || !ts || !l1_nids || !l1_eids || !l1_ts || !reprs || !bitmap || !rank || !out || !key_rows) return TG_EINVAL;
  Carver cv(ws, ws_bytes);
  AttnWs w{};
  if (!carve_attn(m, Q, cv, w)) return TG_EWORKSPACE;
  return attn_forward(m, Q, nids, ts, l1_nids, l1_eids, l1_ts, reprs, bitmap, rank, out, w, as_stream(stream), nullptr,
                      nullptr, nullptr, nullptr, key_rows);
}

extern "C" int tg_linear_fwd(int64_t n, const float* x, int32_t in_f, const tg_linear* lin, int32_t out_f, int32_t relu,
                             float* out, void* stream) {
  if (n < 0 || in_f <= 0 || (in_f % 4) || out_f <= 0 || !lin) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!x || !lin->w || !out) return TG_EINVAL;
  GemmArgs g{};
  g.m_cap = n; g.n = out_f; g.k = in_f; g.a0 = ASeg{x, in_f, in_f, nullptr};
  g.w = lin->w; g.ldw = in_f; g.bias = lin->b; g.c = out; g.ldc = out_f; g.relu = relu; g.alpha = 1.f; g.nbatch = 1;
  return gemm_launch(g, as_stream(stream));
}

// torch.nn.Linear backward on dense rows (the operator path under autograd: MergeLayer, message functions)
extern "C" size_t tg_linear_bwd_workspace_bytes(int32_t in_f, int32_t out_f) {
  if (in_f <= 0 || out_f <= 0) return 0;
  return (size_t)16 * out_f * ((size_t)in_f + 1) * sizeof(float) + 256;  // at most 16 split partials of [out_f, in_f + 1]
}
extern "C" int tg_linear_bwd(int64_t n, const float* x, int32_t in_f, const float* w, int32_t out_f, const float* dy, float* dx,
                             float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
  if (n < 0 || in_f <= 0 || (in_f % 4) || out_f <= 0 || (out_f % 4)) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!dy || (dx && !w) || (dw && !x) || ((dw || db) && !ws)) return TG_EINVAL;
  hipStream_t st = as_stream(stream);
  int rc;
  if (dx) {  // dx = dy W
    GemmArgs g{};
    g.m_cap = n; g.n = in_f; g.k = out_f; g.a0 = ASeg{dy, out_f, out_f, nullptr};
    g.w = w; g.ldw = in_f; g.w_kmajor = 1; g.c = dx; g.ldc = in_f; g.alpha = 1.f; g.nbatch = 1;
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  }
  if (dw) {  // dw = dy^T x, db = column sums of dy (same pass)
    if (ws_bytes < tg_linear_bwd_workspace_bytes(in_f, out_f)) return TG_EWORKSPACE;
    TnArgs tn{};
    tn.m_cap = n; tn.n = out_f; tn.k = in_f; tn.y = dy; tn.ldy = out_f; tn.x0 = ASeg{x, in_f, in_f, nullptr};
    tn.out = dw; tn.ldo = in_f; tn.alpha = 1.f; tn.accumulate = 0; tn.nbatch = 1;
    tn.part = reinterpret_cast<float*>(ws); tn.part_floats = ws_bytes / sizeof(float);
    tn.bias_out = db; tn.bias_accumulate = 0;
    if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
  } else if (db) {
    if ((rc = colsum_launch(n, nullptr, out_f, dy, out_f, 1.f, db, 0, reinterpret_cast<float*>(ws), ws_bytes / sizeof(float), st)) != TG_OK)
      return rc;
  }
  return check_launch("tg_linear_bwd");
}

extern "C" int tg_gru_fwd(int64_t n, const float* x, int32_t xw, const float* h, int32_t d, const float* w_ih,
                          const float* w_hh, const float* b_ih, const float* b_hh, float* out, void* stream) {
  if (n < 0 || d <= 0 || (d % 4) || xw <= 0 || (xw % 4)) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!x || !h || !w_ih || !w_hh || !b_ih || !b_hh || !out) return TG_EINVAL;
  GruArgs a{};
  a.cap = n; a.d = d; a.xw = xw; a.x = ASeg{x, xw, xw, nullptr}; a.h = ASeg{h, d, d, nullptr};
  a.w_ih = w_ih; a.w_hh = w_hh; a.b_ih = b_ih; a.b_hh = b_hh; a.out = out; a.ldo = d;
  return gru_launch(a, as_stream(stream));
}

extern "C" size_t tg_apply_messages_workspace_bytes(const tg_model* m, int64_t cap) {
  if (!attn_dims_ok(m) || cap < 0) return 0;
  return apply_ws_bytes(m, cap);
}

extern "C" int tg_apply_messages(const tg_model* m, const int64_t* outdated, const int32_t* out_pos,
                                 const int32_t* n_outdated, int64_t cap, float* reprs, uint32_t* err, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (!attn_dims_ok(m) || cap < 0) return TG_EINVAL;
  if (cap == 0) return TG_OK;
  if (!outdated || !out_pos || !n_outdated || !reprs || !err) return TG_EINVAL;
  return apply_messages(m, outdated, out_pos, n_outdated, cap, reprs, err, ws, ws_bytes, as_stream(stream));
}

// ---- eager query rows (tiger_hip.h: tg_model.g_table) -------------------------------------------------------------
namespace tg {
__global__ void k_ids32(int64_t n, const int64_t* __restrict__ ids, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int32_t)ids[i];
}
// G rows of the nodes nids[0 .. min(cap, *n_dev)) into m->g_table: c = e(v) + nfeat(v) as the attention centres read it
// (into `crows`, cap x d floats), then the same product the forward pass runs, scattered to the nodes' table rows
// rows of a compact [n, d] buffer -> rows ids[i] of a table (the centre rows of a rebuild into tg_model.c_table)
__global__ void __launch_bounds__(256) k_scatter_rows(int64_t cap, const int32_t* __restrict__ n_dev, int d4,
                                                      const int64_t* __restrict__ ids, const float4* __restrict__ rows,
                                                      float4* __restrict__ table) {
  const int64_t n = n_dev ? min((int64_t)*n_dev, cap) : cap;
  const int64_t total = n * d4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / d4;
    table[ids[i] * d4 + (t - i * d4)] = rows[t];
  }
}

int gtab_rows(const tg_model* m, int64_t cap, const int64_t* nids, const int32_t* rows32, const int32_t* n_dev, float* crows,
              hipStream_t st, bool crows_ready, const CollateRider* collate, bool* rode, int64_t rows_hint) {
  if (rode) *rode = false;
  if (!m->g_table || !m->attn_fused || !m->pending_vals) return TG_EINVAL;
  if (m->row_of) return TG_EUNSUPPORTED;  // (rows32 are node ids)
  const int d = m->d;
  const FusedView f = fused_view(m, m->attn_fused);
  if (!crows_ready) {  // (the GRU updater writes these rows itself, GruArgs.out2 - into the per-node table when there is one)
    hipLaunchKernelGGL(k_attn_centres_direct, dim3(flat_grid(cap * (d / 4), 256)), dim3(256), 0, st, *m, cap, nids,
                       (const float4*)m->nfeats, (float4*)crows, DirectArgs{}, PosArgs{});
    if (m->c_table)
      hipLaunchKernelGGL(k_scatter_rows, dim3(flat_grid(cap * (d / 4), 256)), dim3(256), 0, st, cap, n_dev, d / 4, nids,
                         (const float4*)crows, (float4*)m->c_table);
  }
  GemmArgs g{};
  g.m_cap = cap; g.m_dev = n_dev; g.m_hint = rows_hint; g.n = f.nk; g.k = d;
  // the centre rows: this launch's compact copy, or - gathered by node id - the rows of the per-node table
  g.a0 = m->c_table ? ASeg{m->c_table, d, d, nids} : ASeg{crows, d, d, nullptr};
  g.w = f.wqk; g.ldw = d; g.bias = f.gconst; g.c = m->g_table; g.ldc = f.nk; g.c_rows = rows32; g.alpha = 1.f; g.nbatch = 1;
  return gemm_launch(g, st, nullptr, rode, collate);
}
}  // namespace tg

extern "C" int tg_attn_gtab_rows(const tg_model* m, int64_t n, const int64_t* nids, const int32_t* n_dev, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (!attn_dims_ok(m) || n < 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!nids || !ws) return TG_EINVAL;
  Carver cv(ws, ws_bytes);
  float* crows = cv.take<float>((size_t)n * m->d);
  int32_t* rows32 = cv.take<int32_t>((size_t)n);
  if (!cv.ok) return TG_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_ids32, dim3(flat_grid(n, 256)), dim3(256), 0, st, n, nids, rows32);
  const int rc = gtab_rows(m, n, nids, rows32, n_dev, crows, st, false);
  return rc != TG_OK ? rc : check_launch("tg_attn_gtab_rows");
}

// ---------------------------------------------------------------------------------
// Fused streaming step
// ---------------------------------------------------------------------------------
namespace tg {
enum Stage : int {
  ST_QUERIES = 0, ST_SAMPLE, ST_COMPACT, ST_GATHER, ST_UPDATE, ST_ATTN_PREP, ST_ATTN_Q, ST_ATTN_G, ST_ATTN_CORE,
  ST_ATTN_V, ST_ATTN_O, ST_ATTN_FC1, ST_ATTN_FC2, ST_DEDUP, ST_WRITE_RIGHT, ST_STORE_EVENTS, ST_WRITE_LEFT, ST_EAGER,
  ST_GTAB, ST_COUNT
};
static const char* const kStageNames[ST_COUNT] = {
    "zero_flags", "sample_recent_edges", "unique_compact", "gather_right_memory", "apply_messages(gru)",
    "attn_centres+qconst", "attn_gemm_q", "attn_gemm_g", "attn_core(gather+softmax)", "attn_gemm_v", "attn_gemm_out",
    "attn_gemm_fc1", "attn_gemm_fc2", "dedup_positive", "writeback_phase0", "restarter_targets", "writeback_phase1",
    "eager_updater(gru)", "eager_query_rows(G)"};
static_assert(ST_ATTN_PREP == ST_ATTN_FIRST, "attention stage numbering");
}  // namespace tg

struct tg_profiler {
  hipEvent_t ev[tg::ST_COUNT + 1];
  bool armed;
  tg::KTimer kt;
};
namespace tg {
thread_local KTimer* g_kt = nullptr;
thread_local int g_kt_slot = KT_NONE;
static const char* const kSlotNames[KT_COUNT] = {"collate(sampler+centres)", "attn_core", "fc1", "fc2", "updater", "query_rows",
                                                 "writeback", "gather"};
}  // namespace tg

static inline void prof_mark(tg_profiler* p, int i, hipStream_t st) {
  if (p) (void)hipEventRecord(p->ev[i], st);
}

extern "C" tg_profiler* tg_profiler_create(void) {
  tg_profiler* p = new tg_profiler();
  p->armed = false;
  for (int i = 0; i <= ST_COUNT; ++i)
    if (hipEventCreate(&p->ev[i]) != hipSuccess) {
      delete p;
      return nullptr;
    }
  for (int i = 0; i < KT_COUNT; ++i) {
    p->kt.name[i] = "";
    p->kt.hit[i] = false;
    for (int j = 0; j < 2; ++j)
      if (hipEventCreate(&p->kt.ev[i][j]) != hipSuccess) {
        delete p;
        return nullptr;
      }
  }
  return p;
}
extern "C" void tg_profiler_destroy(tg_profiler* p) {
  if (!p) return;
  for (int i = 0; i <= ST_COUNT; ++i) (void)hipEventDestroy(p->ev[i]);
  for (int i = 0; i < KT_COUNT; ++i)
    for (int j = 0; j < 2; ++j) (void)hipEventDestroy(p->kt.ev[i][j]);
  delete p;
}
extern "C" int tg_profiler_num_stages(void) { return ST_COUNT; }
extern "C" const char* tg_profiler_stage_name(int stage) {
  return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : "";
}
extern "C" int tg_profiler_read(tg_profiler* p, float* ms_out) {
  if (!p || !ms_out || !p->armed) return TG_EINVAL;
  hipError_t e = hipEventSynchronize(p->ev[ST_COUNT]);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_profiler_read");
    return TG_EHIP;
  }
  for (int i = 0; i < ST_COUNT; ++i) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]);
    ms_out[i] = ms;
  }
  return TG_OK;
}

extern "C" int tg_profiler_num_kernel_slots(void) { return KT_COUNT; }
extern "C" const char* tg_profiler_kernel_slot_name(int slot) { return (slot >= 0 && slot < KT_COUNT) ? kSlotNames[slot] : ""; }
// kernel-bound durations of the last profiled step: ms_out[slot] (< 0: no launch was made under the slot),
// names_out[slot] (nullable) = the launch expression of the timed kernel
extern "C" int tg_profiler_kernel_ms(tg_profiler* p, float* ms_out, const char** names_out) {
  if (!p || !ms_out || !p->armed) return TG_EINVAL;
  hipError_t e = hipEventSynchronize(p->ev[ST_COUNT]);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_profiler_kernel_ms");
    return TG_EHIP;
  }
  for (int i = 0; i < KT_COUNT; ++i) {
    float ms = -1.f;
    if (p->kt.hit[i] && hipEventElapsedTime(&ms, p->kt.ev[i][0], p->kt.ev[i][1]) != hipSuccess) ms = -1.f;
    ms_out[i] = ms;
    if (names_out) names_out[i] = p->kt.hit[i] ? p->kt.name[i] : "";
  }
  return TG_OK;
}

namespace tg {
// capacity of the involved-node lists: every slot of the computation graph, or - two layers, where that is Q (1 + K + K^2)
// slots - every node id if that is fewer
static int64_t involved_cap(const tg_model* m, int64_t B, int n_layers) {
  const int64_t Q = 3 * B, K = m->n_neighbors;
  return n_layers == 2 ? std::min<int64_t>(Q * (1 + K + K * K), std::max<int64_t>(m->n_nodes, 1)) : Q * (K + 1);
}

bool carve_step(const tg_model* m, int64_t B, Carver& cv, StepWs& w, int n_layers) {
  const int64_t Q = 3 * B, K = m->n_neighbors, cap = involved_cap(m, B, n_layers);
  const int64_t W = (m->n_nodes + 63) / 64;
  char* z0 = cv.p;
  w.flags = cv.take<uint8_t>((size_t)W * 64);
  w.best = cv.take<unsigned long long>((size_t)cap);
  w.counts = cv.take<int32_t>(8);
  w.zero_bytes = cv.ok ? (size_t)(cv.p - z0) : 0;
  w.bm = cv.take<uint64_t>((size_t)W);
  w.rank = cv.take<uint32_t>((size_t)W + 1);
  w.rank_out = cv.take<uint32_t>((size_t)W + 1);
  w.upos32 = cv.take<int32_t>((size_t)2 * B);
  w.win_row = cv.take<int32_t>((size_t)2 * B);
  w.snap = cv.take<float>((size_t)2 * B * m->d);
  w.snap_ts = cv.take<float>((size_t)2 * B);
  w.nids3 = cv.take<int64_t>((size_t)Q);
  w.eids = cv.take<int64_t>((size_t)B);
  w.ts3 = cv.take<double>((size_t)Q);
  w.ts3f = cv.take<float>((size_t)Q);
  w.l1_nids = cv.take<int64_t>((size_t)Q * K);
  w.l1_eids = cv.take<int64_t>((size_t)Q * K);
  w.l1_ts = cv.take<float>((size_t)Q * K);
  w.involved = cv.take<int64_t>((size_t)cap);
  w.outdated = cv.take<int64_t>((size_t)cap);
  w.out_pos = cv.take<int32_t>((size_t)cap);
  w.upos = cv.take<int64_t>((size_t)2 * B);
  w.index = cv.take<int64_t>((size_t)2 * B);
  w.reprs = cv.take<float>((size_t)cap * m->d);
  w.scan_bytes = tg_unique_compact_workspace_bytes(m->n_nodes);
  w.scan_ws = cv.take<char>(w.scan_bytes);
  if (!carve_attn(m, Q, cv, w.attn)) return false;
  w.apply_bytes = apply_ws_bytes(m, cap);
  w.apply_ws = cv.take<char>(w.apply_bytes);
  w.best_id = cv.take<unsigned long long>((size_t)m->n_nodes);  // lean steps: dedup slots indexed by node id (kept zero)
  if (n_layers == 2) {
    const size_t Q2 = (size_t)Q * K;
    w.h2n = cv.take<int64_t>(Q2 * K);
    w.h2e = cv.take<int64_t>(Q2 * K);
    w.h2t = cv.take<float>(Q2 * K);
    w.ts2 = cv.take<float>(Q2);
    w.emb2 = cv.take<float>(Q2 * m->d);
    if (!carve_attn(m, (int64_t)Q2, cv, w.attn2)) return false;
  }
  // the split updater's buffers (TG_GRU_SPLIT, off by default and measured not faster: DESIGN.md s0.1 of round 4) are carved
  // - and the centres pass stores the 2B time-encoding rows into them - only when the knob is set (static per process, so a
  // prefetched collate sees the same choice): 0.5 GB of workspace and 2B x d dead stores per step at C5 shape otherwise
  if (gru_split_knob()) {
    w.snap_te = cv.take<float>((size_t)2 * B * m->d);
    w.oth = cv.take<int64_t>((size_t)2 * B);
    w.weid = cv.take<int64_t>((size_t)2 * B);
    w.gi = cv.take<float>((size_t)2 * B * 3 * m->d);
  }
  return cv.ok;
}
// the carve above on a dry run: workspace size and the must-be-zero prefix without a second copy of the layout
static bool carve_dry(const tg_model* m, int64_t B, int n_layers, size_t* bytes, size_t* zero_bytes) {
  char* const base = reinterpret_cast<char*>((uintptr_t)1 << 20);  // never dereferenced
  Carver cv(base, (size_t)1 << 60);
  StepWs w{};
  if (!carve_step(m, B, cv, w, n_layers)) return false;
  if (bytes) *bytes = (size_t)(cv.p - base);
  if (zero_bytes) *zero_bytes = w.zero_bytes;
  return true;
}
}  // namespace tg

extern "C" size_t tg_stream_step_workspace_bytes2(const tg_model* m, int64_t B, int32_t n_layers);
extern "C" size_t tg_stream_step_workspace_bytes(const tg_model* m, int64_t B) {
  return tg_stream_step_workspace_bytes2(m, B, 1);
}
extern "C" size_t tg_stream_step_workspace_bytes2(const tg_model* m, int64_t B, int32_t n_layers) {
  if (!attn_dims_ok(m) || B <= 0 || m->n_nodes <= 0 || (n_layers != 1 && n_layers != 2)) return 0;
  size_t b = 0;
  return tg::carve_dry(m, B, n_layers, &b, nullptr) ? b + 256 : 0;
}

extern "C" size_t tg_stream_step_zero_bytes2(const tg_model* m, int64_t B, int32_t n_layers) {
  if (!attn_dims_ok(m) || B <= 0 || m->n_nodes <= 0 || (n_layers != 1 && n_layers != 2)) return 0;
  size_t z = 0;
  return tg::carve_dry(m, B, n_layers, nullptr, &z) ? z : 0;
}
extern "C" size_t tg_stream_step_zero_bytes(const tg_model* m, int64_t B) { return tg_stream_step_zero_bytes2(m, B, 1); }

namespace tg {
__global__ void k_slot_times(int64_t n, int K, const float* __restrict__ ts, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = ts[i / K];
}

static WritebackArgs writeback_args(const tg_model* m, const tg_step_io* io, StepWs& w);

// Which form a step takes: one place, shared by step_forward and tg_stream_step_form (the host asks before / after a step
// whether the per-node tables are read and kept by it).
struct StepForm {
  bool direct, fused_wb, lean, gtab, lz_static;
};
static StepForm step_form(const tg_model* m, const tg_step_io* io, bool eager, bool drop, bool inner) {
  StepForm f{};
  const tg_lazy_restart* lz = (io->lazy && !io->embed_only) ? io->lazy : nullptr;
  // eager updates, direct form (default; TG_EAGER_DIRECT=0 keeps the compact copy): centres and neighbour rows are read
  // from pending / right themselves, so there is no gather launch and no reprs buffer
  static const int direct_knob = getenv("TG_EAGER_DIRECT") ? atoi(getenv("TG_EAGER_DIRECT")) : 1;
  f.direct = eager && direct_knob != 0 && !io->eager_copy && !io->collate_only;
  // the one-launch write-back needs the snapshot; the restarter targets (h_prev_*) are read between STEP 4 and STEP 6,
  // so a step that outputs them keeps the two-phase write-back
  f.fused_wb = f.direct && !io->embed_only && !io->h_prev_left && !io->h_prev_right;
  // lean: nothing in such a step needs the involved / outdated sets, so they are not formed (tiger_hip.h, tg_step_io.lean)
  // (an embed-only step has no write-back to clean up after: lean, it touches none of the self-cleaning state at all)
  // With the in-step restart loop of the STATIC restarter a lean step still marks the involved flags (the loop's only
  // input) but forms no sorted set; the list form (any other restarter) keeps the full step
  f.lz_static = lz && !lz->list;
  f.lean = io->lean && f.direct && (!lz || f.lz_static) && (f.fused_wb || io->embed_only);
  // eager query rows: a full eager step of a model that carries the table (it refreshes the table at its end)
  // (with the in-step restart loop: only with the centre-row table, whose rows of the re-initialised nodes the loop's
  // kernel rewrites itself - their query rows are refreshed right behind it)
  f.gtab = eager && m->g_table && m->attn_fused && (!lz || (f.lz_static && m->c_table && f.lean)) && !inner && !io->embed_only &&
           !io->collate_only && !drop;
  return f;
}

int step_forward(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, StepWs& w, float* gates, hipStream_t st,
                 tg_profiler* pf, const DropCfg* drop, bool eager) {
  w.eager = eager;
  w.wb_rode = false;
  w.upd_done = false;
  const tg_model* inner = w.h2n ? io->inner : nullptr;  // two attention layers (the workspace was carved for them)
  const int64_t B = io->B, Q = 3 * B, K = m->n_neighbors, cap = involved_cap(m, B, inner ? 2 : 1);
  prof_mark(pf, ST_QUERIES, st);
  hipError_t e = hipSuccess;
  int rc;
  // ---- collate (data_loader.py:77-131): queries + temporal neighbours + involved flags, one launch
  prof_mark(pf, ST_SAMPLE, st);
  w.l1n = io->l1_nids ? io->l1_nids : w.l1_nids;
  w.l1e = io->l1_eids ? io->l1_eids : w.l1_eids;
  w.l1t = io->l1_ts ? io->l1_ts : w.l1_ts;
  const tg_lazy_restart* lz = (io->lazy && !io->embed_only) ? io->lazy : nullptr;
  const StepForm form = step_form(m, io, eager, drop != nullptr, inner != nullptr);
  w.direct = form.direct;
  w.fused_wb = form.fused_wb;
  w.lean = form.lean;
  const bool need_flags = !w.lean || lz != nullptr;
  const bool untouched = w.lean && io->embed_only;  // no flags, no dedup slots, no counts
  if ((!io->ws_is_clean || io->embed_only || io->collate_only) && !untouched) e = hipMemsetAsync(w.flags, 0, w.zero_bytes, st);
  if (e == hipSuccess && w.lean && !io->embed_only && !io->ws_is_clean)
    e = hipMemsetAsync(w.best_id, 0, (size_t)m->n_nodes * 8, st);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_stream_step memset");
    return TG_EHIP;
  }
  // the positive-node dedup (needed by the write-back) rides on two launches of the forward pass
  PosArgs pos{B, w.nids3, w.ts3f, w.bm, w.rank, w.best, w.counts + 2, w.upos, w.index, w.upos32, nullptr, nullptr};
  if (w.lean) { pos.bm = nullptr; pos.rank = nullptr; pos.best = w.best_id; pos.chk_err = io->err; }
  if (untouched) {  // no dedup (there is no write-back), but the core still checks its neighbours and moves the offset on
    pos.best = nullptr;
    pos.advance_off = (io->offset_dev && io->advance) ? io->offset_dev : nullptr;
  }
  // write-back rider (tg_common.h: WbRider): STEP 4-5 share the launch of the attention block's last product, whose
  // epilogue stores STEP 6's rows; TG_WB_RIDER=0 keeps the write-back launch
  static const int wbr_knob = getenv("TG_WB_RIDER") ? atoi(getenv("TG_WB_RIDER")) : 1;
  // (not with io->h_new: those rows are read from the tables after the attention block, i.e. before STEP 4 must run)
  const bool want_rider = w.fused_wb && wbr_knob != 0 && !drop && !io->h_new;
  if (want_rider) pos.win_row = w.win_row;
  // split updater (tg_dense.h: GruTail; VERDICT r03 task 1a): the updater's input-side product W_ih msg - 80 % of its flops,
  // independent of the attention block - leaves the updater launch.  Needs the raw messages as gathered segments of the
  // snapshot (GRU, no message transform), h = this batch's h(t-) (upd_src = left: fc2's rows), the per-node tables and the
  // write-back rider's winner lists.  TG_GRU_SPLIT: 1 = W_ih msg as a second problem of fc1's launch + the tail (W_hh h
  // through the parameter product W_hh W2, gates) on fc2's launch: NO updater launch; 2 = W_ih msg on fc2's launch + the
  // tail as a short launch behind it.  Both parity-green, both measured NOT faster at C2 (MI355X, 100 replays; 0 / 1 / 2:
  // 83.7-86.4 / 83.7-85.0 / 86.5-87.8 us per step on three boxes): a 48 x 48 x 688 tile costs its launch 11-14 us wherever
  // it rides (fc1 25.0 -> 39.5 us, fc2 13.0 -> 24.2 us by HIP events) against the 16.3 us the updater launch gives back,
  // and one fork / join inside the captured graph costs more than the product (tools/micro/fork_join.py).  Default 0.
  const int split_knob = gru_split_knob();
  const float* tail_w = gru_tail_weights(m);
  const bool split_ok = split_knob != 0 && tail_w && m->upd_fn == TG_UPD_GRU && m->tsfm == TG_TSFM_ID && m->upd_src == TG_SRC_LEFT;
  if (want_rider && split_ok) { pos.eids = w.eids; pos.oth = w.oth; pos.weid = w.weid; }
  const PosArgs* pp = (io->embed_only && !untouched) ? nullptr : &pos;
  w.gtab = form.gtab;
  // collate prefetch (tg_step_io.prefetch_state): this step runs the NEXT batch's sampler + centres on its last launch;
  // `prefetched`: the previous call did that for this batch (a repeated collate would be harmless, just wasted)
  static const int pf_knob = getenv("TG_PREFETCH") ? atoi(getenv("TG_PREFETCH")) : 1;
  w.prefetch = pf_knob != 0 && io->prefetch_state && io->stream_len > 0 && io->offset_dev && io->advance && io->ws_is_clean &&
               w.lean && w.gtab && w.fused_wb && !lz && io->strategy == 0 && K <= 16 && !io->l1_nids && !io->l1_eids && !io->l1_ts &&
               // (large batches: the last product's launch takes no riders, see gemm_launch - the sampler half then runs on
               // the side lane beside the updater, the centres half behind the query rows: step_writeback_b)
               (B <= SIDE_MIN_B || (m->c_table && side_lane(st) != nullptr));
  w.prefetch_side = w.prefetch && B > SIDE_MIN_B;
  const int pf_in = io->prefetch_state ? *io->prefetch_state : 0;
  const bool prefetched = w.prefetch && pf_in == 1;
  if (io->prefetch_state) *io->prefetch_state = 0;  // set again by the end of the step, once the rider is enqueued
  if (pf_in != 0 && !prefetched) {
    // a prefetch that is not used (stale, or this step takes another form): its first dedup pass left maxima in the
    // node-indexed slot table, which only the write-back of THAT batch's lean step would have cleared
    if ((e = hipMemsetAsync(w.best_id, 0, (size_t)m->n_nodes * 8, st)) != hipSuccess) {
      set_hip_error(e, "tg_stream_step memset (discarded prefetch)");
      return TG_EHIP;
    }
  }
  DirectArgs da{w.lean ? nullptr : w.outdated, w.counts + 1, cap, io->err, w.fused_wb ? (float4*)w.snap : nullptr,
                w.snap_ts, 2 * B, w.lean ? 1 : 0};
  if (w.fused_wb && split_ok) da.snap_te = (float4*)w.snap_te;  // (whenever the snapshot is taken: a prefetched collate of
                                                                //  this batch ran before this step chose its form)
  // lean: the centres need nothing the sampler produces and share its launch
  // (with the per-node table of centre rows - tg_model.c_table, a step that uses the query-row table - no per-batch copy
  // of the centre rows is made: the centres pass keeps its checks, the first dedup pass and the snapshot)
  const bool ctab = w.gtab && m->c_table;
  const CentresRider rider{*m, (const float4*)m->nfeats, ctab ? (float4*)nullptr : (float4*)w.attn.cc, da, pp ? pos : PosArgs{},
                           flat_grid((ctab ? 2 * B : Q) * (m->d / 4), 256)};
  w.pos_args = pp ? pos : PosArgs{};
  w.da_args = da;
  const bool recent_nodes = io->strategy == 1, uniform = io->strategy == 2;
  if (io->strategy < 0 || io->strategy > 2 || (uniform && !io->mt_state)) return TG_EUNSUPPORTED;
  if (prefetched) {
    // sampler, centres, first dedup pass and snapshot of this batch rode on the previous step's last launch
  } else if (recent_nodes || uniform) {  // query arrays first, then the sampler of graph.py:129-143 / :101-115 and the involved flags
    // (the lazy-restart loop with `uniform`: a collate-only pass would consume draws of the graph's stream the step repeats)
    if (lz && uniform) return TG_EUNSUPPORTED;
    hipLaunchKernelGGL(k_build_queries, dim3(flat_grid(Q, 256)), dim3(256), 0, st, B, io->src, io->dst, io->neg, io->ts,
                       io->eids, (const int64_t*)io->offset_dev, w.nids3, w.ts3, w.ts3f, w.eids,
                       lz ? reinterpret_cast<uint32_t*>(w.counts + 4) : nullptr);
    // uniform: the graph's MT19937 stream is consumed per non-empty query, in query order (src, dst, neg of the batch, as
    // data_loader.py:79-81 concatenates them): one wavefront walks the queries, 64 words of the stream at a time
    if ((rc = uniform ? sample_uniform_launch(g, Q, w.nids3, w.ts3, (int32_t)K, io->mt_state, w.l1n, w.l1e, w.l1t,
                                              need_flags ? w.flags : nullptr, st)
                      : sample_nodes_launch(g, Q, w.nids3, w.ts3, (int32_t)K, w.l1n, w.l1e, w.l1t, need_flags ? w.flags : nullptr,
                                            st)) != TG_OK)
      return rc;
  } else if (KSlot ks_(KT_COLLATE); (rc = sample_batch_launch(g, B, io->src, io->dst, io->neg, io->ts, io->eids, io->offset_dev, (int32_t)K,
                                       w.nids3, w.ts3f, w.eids, w.l1n, w.l1e, w.l1t, need_flags ? w.flags : nullptr, st,
                                       lz ? reinterpret_cast<uint32_t*>(w.counts + 4) : nullptr,
                                       (w.lean && !lz) ? &rider : nullptr)) != TG_OK)  // (centres: behind the restart loop)
    return rc;
  // a caller that runs work beside the rest of the forward pass (the training step's restarter: it needs the id list only)
  if (w.collate_done && hipEventRecord(w.collate_done, st) == hipSuccess) w.collate_recorded = true;
  if (io->dbg_l1_nids || io->dbg_l1_eids || io->dbg_l1_ts) {  // the lists this step consumes, before its last launch prefetches the next
    const size_t n = (size_t)Q * K;
    if (io->dbg_l1_nids && (e = hipMemcpyAsync(io->dbg_l1_nids, w.l1n, n * 8, hipMemcpyDeviceToDevice, st)) != hipSuccess) return TG_EHIP;
    if (io->dbg_l1_eids && (e = hipMemcpyAsync(io->dbg_l1_eids, w.l1e, n * 8, hipMemcpyDeviceToDevice, st)) != hipSuccess) return TG_EHIP;
    if (io->dbg_l1_ts && (e = hipMemcpyAsync(io->dbg_l1_ts, w.l1t, n * 4, hipMemcpyDeviceToDevice, st)) != hipSuccess) return TG_EHIP;
  }
  // second hop (data_loader.py:128-131): every neighbour slot (padding included) queried at its own float32 timestamp
  // - with the graph's own strategy (uniform: the graph's stream goes on behind the first hop's draws)
  if (inner && (rc = uniform ? sample_uniform_f32_launch(g, Q * K, w.l1n, w.l1t, (int32_t)K, io->mt_state, w.h2n, w.h2e, w.h2t,
                                                         need_flags ? w.flags : nullptr, st)
                             : (recent_nodes ? sample_nodes_f32_launch : sample_edges_f32_launch)(
                                   g, Q * K, w.l1n, w.l1t, (int32_t)K, w.h2n, w.h2e, w.h2t, need_flags ? w.flags : nullptr, st)) !=
                   TG_OK)
    return rc;
  // lazy restart (train_self_supervised.py:152-163): before STEP 1, because a restarted node loses its pending message
  const bool lz_tables = lz && w.gtab;  // the loop also keeps the per-node tables current (lists what it re-initialised)
  if (lz && (rc = lazy_restart_launch(g, m, lz, w.flags, reinterpret_cast<const uint32_t*>(w.counts + 4), w.counts + 3, st,
                                      lz_tables ? w.outdated : nullptr, lz_tables ? w.out_pos : nullptr)) != TG_OK)
    return rc;
  // ... their query rows: one product over the listed nodes (usually none or a handful; everybody involved right after
  // a trigger).  Nodes that lost their message at a trigger and are not involved keep stale rows - they are not up to
  // date either, so they are re-initialised (and refreshed here) before any batch reads them
  if (lz_tables && (rc = gtab_rows(m, cap, w.outdated, w.out_pos, w.counts + 3, nullptr, st, true, nullptr, nullptr, 1024)) != TG_OK)
    return rc;
  prof_mark(pf, ST_COMPACT, st);
  w.inv = io->involved ? io->involved : w.involved;
  // involved = sorted(set(...)); outdated = involved & has-message (memory.py:108-126)
  if (!w.lean &&
      (rc = unique_compact_launch(w.flags, w.bm, m->n_nodes, w.rank, w.inv, w.counts + 0, cap,
                                  m->row_of ? nullptr : m->has_msg,  // (row-indexed bitmap: a collate-only pass has no use for it)
                                  w.rank_out, w.outdated, w.out_pos, w.counts + 1, w.scan_ws, w.scan_bytes, st)) != TG_OK)
    return rc;
  if (io->collate_only) return check_launch("tg_stream_step(collate_only)");
  prof_mark(pf, ST_GATHER, st);
  w.dedup_done = pp != nullptr && pp->best != nullptr;
  // ---- STEP 1-2: reprs = right_memory[involved] (+ invariants); outdated rows <- updater(...)
  if (KSlot ks_(KT_GATHER); !w.direct &&
      (rc = consume_gather_check_launch(m, w.inv, w.counts + 0, cap, w.reprs, w.outdated, w.counts + 1, io->err, st, pp,
                                        eager)) != TG_OK)
    return rc;
  prof_mark(pf, ST_UPDATE, st);
  if (KSlot ks_(KT_UPDATER); !eager &&  // eager: the rows were gathered from the table of precomputed updater rows just now
      (rc = apply_messages(m, w.outdated, w.out_pos, w.counts + 1, cap, w.reprs, io->err, w.apply_ws, w.apply_bytes,
                           st, true, gates, io->rows_hint)) != TG_OK)
    return rc;
  // ---- STEP 3: temporal embeddings of cat[src, dst, neg]
  const float* key_rows = nullptr;
  if (inner) {
    // the Q*K neighbour slots embedded with the second layer over their own neighbours, at the ROOT's query time
    // (temporal_agg_modules.py:57-66); padding slots (node 0) are embedded too and masked by the first layer
    const int64_t Q2 = Q * K;
    hipLaunchKernelGGL(k_slot_times, dim3(flat_grid(Q2, 256)), dim3(256), 0, st, Q2, (int)K, (const float*)w.ts3f, w.ts2);
    const DirectArgs da2{nullptr, w.counts + 1, cap, io->err, nullptr, nullptr, 0, w.lean ? 1 : 0};
    PosArgs pos2{};  // no dedup rides on the inner launches; lean: the neighbours' time invariants still do
    pos2.chk_err = w.lean ? io->err : nullptr;
    if ((rc = attn_forward(inner, Q2, w.l1n, w.ts2, w.h2n, w.h2e, w.h2t, w.reprs, w.bm, w.rank, w.emb2, w.attn2, st, nullptr,
                           drop, w.direct ? &pos2 : nullptr, w.direct ? &da2 : nullptr, nullptr, false)) != TG_OK)
      return rc;
    key_rows = w.emb2;
  }
  WbRider wbr{};
  if (want_rider) {
    wbr.m = *m;
    wbr.a = writeback_args(m, io, w);
    wbr.a.snap = w.snap;
    wbr.a.snap_ts = w.snap_ts;
  }
  GruSplit gs{};
  gs.variant = split_knob;
  const bool gsplit = split_knob != 0 && split_ok && want_rider && pp && w.gtab && m->c_table && !lz && !inner && !key_rows;
  if (gsplit) {
    const int d = m->d, mw = 3 * d + m->d_e;
    GemmArgs& gi = gs.gi;  // gi = [snap[index] | snap[oth] | efeat[weid] | snap_te[index]] W_ih^T + b_ih over the winners
    gi.m_cap = 2 * B; gi.m_dev = w.counts + 2; gi.m_hint = io->rows_hint; gi.n = 3 * d; gi.k = mw;
    gi.a0 = ASeg{w.snap, d, d, w.index};
    gi.a1 = ASeg{w.snap, d, d, w.oth};
    gi.a2 = ASeg{m->efeats, m->d_e, m->d_e, w.weid};  // (no edge table: a slice of zeros)
    gi.a3 = ASeg{w.snap_te, d, d, w.index};
    gi.w = m->gru_w_ih; gi.ldw = mw; gi.bias = m->gru_b_ih; gi.c = w.gi; gi.ldc = 3 * d; gi.alpha = 1.f; gi.nbatch = 1;
    GruTail& t = gs.tail;
    t.cap = 2 * B; t.n_dev = w.counts + 2; t.d = d; t.t = w.attn.t; t.t_rows = w.index;
    t.w = tail_w; t.b = tail_w + (size_t)4 * d * d; t.gi = w.gi;
    t.out = m->pending_vals; t.out_rows = w.upos32; t.out2 = m->c_table; t.add2 = m->nfeats; t.rows_hint = io->rows_hint;
    if (gs.variant == 2) {  // the tail behind fc2, reading what the reference reads: left[v] and weight_hh / bias_hh
      t.direct = 1; t.t = m->left_vals; t.t_rows = w.upos; t.w = m->gru_w_hh; t.b = m->gru_b_hh;
      // ... and W_ih msg over the mailbox rows the write-back rider has just stored (on fc1's launch)
      static const int box_knob = getenv("TG_GRU_SPLIT_BOX") ? atoi(getenv("TG_GRU_SPLIT_BOX")) : 1;
      if (box_knob) {
        gi.a0 = ASeg{m->msg_vals, mw, mw, w.upos};
        gi.a1 = gi.a2 = gi.a3 = ASeg{};
      }
    }
  }
  // collate prefetch: TG_PREFETCH_SPLIT=1 lets the sampler half of the NEXT batch's collate share fc2's launch (it reads the
  // graph and the stream only) and leaves the centres on the step's last launch.  Parity-green, measured SLOWER at C2 (87.2
  // against 85.2 us per step on the same box: fc2's launch grows by more than the last launch gives back); default 0
  static const int pfs_knob = getenv("TG_PREFETCH_SPLIT") ? atoi(getenv("TG_PREFETCH_SPLIT")) : 0;
  CollateRider co_s{};
  if (w.prefetch && pfs_knob != 0 && !gsplit) {
    co_s.s = SampleBatchArgs{*g, io->B, io->src, io->dst, io->neg, io->ts, io->eids, (const int64_t*)io->offset_dev,
                             (int)m->n_neighbors, w.nids3, w.ts3f, w.eids, w.l1n, w.l1e, w.l1t, nullptr, nullptr};
    co_s.cr = CentresRider{*m, (const float4*)m->nfeats, nullptr, DirectArgs{}, PosArgs{}, 0u};
    co_s.stream_len = io->stream_len;
  }
  w.sampler_rode = false;
  if ((rc = attn_forward(m, Q, w.nids3, w.ts3f, w.l1n, w.l1e, w.l1t, w.reprs, w.bm, w.rank, io->h, w.attn, st, pf,
                         drop, pp, w.direct ? &da : nullptr, key_rows, w.lean && io->strategy == 0 && !lz, w.gtab,
                         want_rider ? &wbr : w.ext_rider, &w.wb_rode, gsplit ? &gs : nullptr,
                         (w.prefetch && pfs_knob != 0 && !gsplit) ? &co_s : nullptr, &w.sampler_rode)) != TG_OK)
    return rc;
  w.upd_done = gs.done;
  w.tail_pending = gsplit && gs.variant == 2 && gs.gi_done;
  if (w.tail_pending) w.tail = gs.tail;
  prof_mark(pf, ST_DEDUP, st);
  if (io->h_new) {  // h(t'+) rows of cat[src, dst]: reprs[local(node)], or the table rows themselves
    if (w.direct)
      hipLaunchKernelGGL(k_attn_centres_direct, dim3(flat_grid(2 * B * (m->d / 4), 256)), dim3(256), 0, st, *m, 2 * B,
                         w.nids3, (const float4*)nullptr, (float4*)io->h_new, DirectArgs{}, PosArgs{});
    else
      hipLaunchKernelGGL(k_attn_centres, dim3(flat_grid(2 * B * (m->d / 4), 256)), dim3(256), 0, st, 2 * B, m->d / 4,
                         w.nids3, (const float4*)w.reprs, w.bm, w.rank, (const float4*)nullptr, (float4*)io->h_new, 0,
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                         (float*)nullptr, PosArgs{});
  }
  return check_launch("tg_stream_step(forward)");
}

// ---- dedup of positive nodes (select_latest_nids on float32 ts, tiger.py:232,419; memory.py:98),
// STEP 4-5 and the restarter targets.  STEP 4-6 (tiger.py:229-255) take two launches; STEP 5
// shares a launch with whichever of STEP 4 / STEP 6 does not write the message memory
// (see tg_memory.hip)
static WritebackArgs writeback_args(const tg_model* m, const tg_step_io* io, StepWs& w) {
  const int64_t B = io->B;
  WritebackArgs wa{};
  wa.B = B; wa.src = w.nids3; wa.dst = w.nids3 + B; wa.eids = w.eids; wa.upos = w.upos; wa.index = w.index;
  wa.ts = w.ts3f;
  wa.n_upos = w.counts + 2; wa.reprs = w.reprs; wa.bm = w.bm; wa.rank = w.rank; wa.h = io->h; wa.err = io->err;
  wa.counts_src = io->counts ? w.counts : nullptr; wa.counts_dst = io->counts;
  wa.offset_dev = (io->offset_dev && io->advance) ? io->offset_dev : nullptr;
  wa.clean_flags = (w.lean && !io->lazy) ? nullptr : w.flags; wa.flag_bytes = (int64_t)((m->n_nodes + 63) / 64) * 64;
  wa.clean_best = w.lean ? w.best_id : w.best; wa.clean_best_by_pos = w.lean ? 1 : 0;
  wa.clean_counts = w.counts;
  wa.lazy_batch = (io->lazy && io->lazy->batch_dev) ? io->lazy->batch_dev : nullptr;
  wa.new_from_pending = w.direct ? 1 : 0;  // no reprs copy was made: STEP 4 reads the owner table of updater rows
  return wa;
}

int step_writeback_a(const tg_model* m, const tg_step_io* io, StepWs& w, hipStream_t st, tg_profiler* pf) {
  const int64_t B = io->B;
  if (w.fused_wb) {  // STEP 4-6 run as ONE launch from step_writeback_b
    prof_mark(pf, ST_WRITE_RIGHT, st);
    prof_mark(pf, ST_STORE_EVENTS, st);
    return TG_OK;
  }
  const int64_t* src = w.nids3;
  const int64_t* dst = w.nids3 + B;
  if (!w.dedup_done) {
    hipLaunchKernelGGL(k_pos_max, dim3(flat_grid(2 * B, 256)), dim3(256), 0, st, B, src, dst, w.ts3f, w.bm, w.rank,
                       w.best, w.counts + 2);
    hipLaunchKernelGGL(k_pos_winners, dim3(flat_grid(2 * B, 256)), dim3(256), 0, st, B, src, dst, w.ts3f, w.bm, w.rank,
                       w.best, w.upos, w.index, w.counts + 2);
  }
  const WritebackArgs wa = writeback_args(m, io, w);
  int rc;
  prof_mark(pf, ST_WRITE_RIGHT, st);
  if ((rc = writeback_launch(m, wa, 0, st)) != TG_OK) return rc;
  prof_mark(pf, ST_STORE_EVENTS, st);
  // ---- side outputs for the restarter (tiger.py:248-251): after STEP 4, before STEP 6
  if (io->h_prev_left) {
    if ((rc = tg_gather_rows(2 * B, w.nids3, m->d, m->left_vals, io->h_prev_left, nullptr, nullptr, (void*)st)) != TG_OK)
      return rc;
  }
  if (io->h_prev_right) {
    if ((rc = tg_gather_rows(2 * B, w.nids3, m->d, m->right_vals, io->h_prev_right, nullptr, nullptr, (void*)st)) != TG_OK)
      return rc;
  }
  return TG_OK;
}

// the collate part of a batch as a launch of its own (collate prefetch when the last product's kernel hosts no rider)
__global__ void __launch_bounds__(256) k_collate(CollateRider co) { co.run(blockIdx.x); }
static void collate_blocks_standalone(CollateRider& c) {
  const int64_t Q = 3 * c.s.B;
  c.sblocks = c.parts == 2 ? 0u : flat_grid(Q, 16);
  c.cr.blocks = c.parts == 1 ? 0u : flat_grid(Q * (c.cr.m.d / 4), 256);
  c.blocks = c.sblocks + c.cr.blocks;
}

int step_writeback_b(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, StepWs& w, hipStream_t st, tg_profiler* pf) {
  // unique positive nodes of the batch: slot 2 of the counts, or - one-pass write-back - its copy (see writeback_fused_body)
  const int32_t* n_upos = w.fused_wb ? w.counts + 5 : w.counts + 2;
  WritebackArgs wa = writeback_args(m, io, w);
  prof_mark(pf, ST_WRITE_LEFT, st);
  int rc;
  CollateRider co{};
  SideLane* lane = nullptr;
  if (w.fused_wb) {
    wa.snap = w.snap;
    wa.snap_ts = w.snap_ts;
  }
  // (rider: STEP 4-6 already ran inside the launch of the attention block's last product)
  if (KSlot ks_(KT_WRITEBACK); !(w.fused_wb && w.wb_rode) && (rc = writeback_launch(m, wa, w.fused_wb ? 2 : 1, st)) != TG_OK) return rc;
  prof_mark(pf, ST_EAGER, st);
  if (w.eager && !io->embed_only) {
    // every unique positive node has just received a message (STEP 5) and its memories are final for this batch
    // (STEP 4 / STEP 6): the row a later batch would compute on the fly when the node turns up as a neighbour,
    // pending[v] = updater(upd_memory[v], tsfm(mailbox[v])), is computed here, once
    const int64_t P = 2 * io->B;
    // rows_hint (eager steps): the caller's bound on the unique positive nodes of a batch; performance only
    const int64_t bound = io->rows_hint > 0 ? std::min<int64_t>(P, io->rows_hint) : P;
    // With eager query rows the GRU epilogue also leaves the attention-centre form of its rows (h + node features) in the
    // centre-row buffer of the forward pass, which is free again
    // (centre-row buffer of these launches: the block's fc1 output buffer, free since fc2 - NOT the centre rows of the
    // forward pass, which the prefetched centres of the next batch overwrite during the query-row launch)
    const bool cr = w.gtab && m->upd_fn == TG_UPD_GRU;
    if (w.prefetch) {  // the next batch's collate part rides on the step's last launch (tg_sample.h: CollateRider)
      co.s = SampleBatchArgs{*g, io->B, io->src, io->dst, io->neg, io->ts, io->eids, (const int64_t*)io->offset_dev,
                             (int)m->n_neighbors, w.nids3, w.ts3f, w.eids, w.l1n, w.l1e, w.l1t, nullptr, nullptr};
      co.cr = CentresRider{*m, (const float4*)m->nfeats, m->c_table ? (float4*)nullptr : (float4*)w.attn.cc, w.da_args,
                           w.pos_args, 0u};
      co.stream_len = io->stream_len;
      co.parts = w.sampler_rode ? 2u : 0u;  // (the sampler half rode on fc2's launch already)
    }
    const bool ctab = cr && m->c_table;  // ... or straight into the per-node table of centre rows
    if (w.prefetch_side && (lane = side_lane(st)) != nullptr) {
      // the NEXT batch's sampler (graph + stream only) beside the updater and the query rows; it writes the step's query
      // arrays and neighbour lists, which nothing from here on reads (as with TG_PREFETCH_SPLIT)
      if (!lane_fork(lane, 1, st)) return TG_EHIP;
      CollateRider cs = co;
      cs.parts = 1u;
      collate_blocks_standalone(cs);
      hipLaunchKernelGGL(k_collate, dim3(cs.blocks), dim3(256), 0, lane->s, cs);
      co.parts = 2u;  // the centres half (reads the state the updater is about to finish, and the sampler's query ids)
    }
    KSlot ks_upd(KT_UPDATER);
    if (w.tail_pending && (rc = gru_tail_launch(w.tail, st)) != TG_OK) return rc;  // (split updater, variant 2)
    if (!w.upd_done && !w.tail_pending &&  // (split updater: these rows were finished on fc2's launch)
        (rc = apply_messages(m, w.upos, w.upos32, n_upos, P, m->pending_vals, io->err, w.apply_ws, w.apply_bytes, st,
                             true, nullptr, bound, cr ? (ctab ? m->c_table : w.attn.t) : nullptr, cr ? m->nfeats : nullptr,
                             ctab)) != TG_OK)
      return rc;
  }
  prof_mark(pf, ST_GTAB, st);
  if (w.gtab) {
    // ... and the query rows of the same nodes: their effective rows have just changed (tg_model.g_table)
    bool rode = false;
    KSlot ks_q(KT_QROWS);
    if ((rc = gtab_rows(m, 2 * io->B, w.upos, w.upos32, n_upos, w.attn.t, st, m->upd_fn == TG_UPD_GRU,
                        (w.prefetch && !lane) ? &co : nullptr, &rode, io->rows_hint)) != TG_OK)  // (lane: the centres half waits for the join)
      return rc;
    if (lane && !lane_join(lane, 1, st)) return TG_EHIP;
    if (w.prefetch && !rode) {  // this product's kernel does not host riders: the same work as a launch of its own
      collate_blocks_standalone(co);
      hipLaunchKernelGGL(k_collate, dim3(co.blocks), dim3(256), 0, st, co);
    }
    if (w.prefetch) *io->prefetch_state = 1;
  }
  prof_mark(pf, ST_COUNT, st);
  if (pf) pf->armed = true;
  return check_launch("tg_stream_step");
}
}  // namespace tg

extern "C" int32_t tg_stream_step_form(const tg_model* m, const tg_step_io* io) {
  if (!m || !io) return 0;
  const tg::StepForm f = tg::step_form(m, io, m->pending_vals != nullptr, false, io->inner != nullptr);
  return (f.direct ? TG_FORM_DIRECT : 0) | (f.fused_wb ? TG_FORM_FUSED_WB : 0) | (f.lean ? TG_FORM_LEAN : 0) |
         (f.gtab ? TG_FORM_TABLES : 0);
}

namespace tg {
int stream_step_ext(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, void* ws, size_t ws_bytes, hipStream_t st,
                    const WbRider* ext_rider, bool* ext_rode) {
  if (ext_rode) *ext_rode = false;
  if (!attn_dims_ok(m) || !g || !io || io->B <= 0) return TG_EINVAL;
  if (!io->src || !io->dst || !io->neg || !io->ts || !io->eids || (!io->h && !io->collate_only) || !io->err)
    return TG_EINVAL;
  if (g->num_node != m->n_nodes) return TG_EINVAL;
  // physically partitioned state (tg_model.row_of): only the forms that address state by row
  if (m->row_of && !io->collate_only && !(io->embed_only && io->lean && m->pending_vals && !io->inner)) return TG_EUNSUPPORTED;
  tg_profiler* pf = (tg_profiler*)io->profiler;
  struct KtScope {  // kernel-bound timing of the step's main launches while a profiler is attached (tg_common.h)
    explicit KtScope(tg_profiler* p) {
      if (p)
        for (int i = 0; i < KT_COUNT; ++i) p->kt.hit[i] = false;
      g_kt = p ? &p->kt : nullptr;
    }
    ~KtScope() { g_kt = nullptr; }
  } kt_scope(pf);
  Carver cv(ws, ws_bytes);
  StepWs w{};
  if (!carve_step(m, io->B, cv, w, io->inner ? 2 : 1)) return TG_EWORKSPACE;
  w.ext_rider = ext_rider;
  int rc;
  // eager updates need the full step (the updater launch at its end keeps the table current)
  // embed_only still GATHERS the precomputed rows when the table is there (the partitioned multi-GPU path keeps it
  // current through tg_apply_messages after its own write-back); only a full step runs the updater at its end
  const bool eager = m->pending_vals != nullptr;
  if ((rc = step_forward(m, g, io, w, nullptr, st, pf, nullptr, eager)) != TG_OK) return rc;
  if (ext_rode) *ext_rode = ext_rider && w.wb_rode;
  if (pf && (io->embed_only || io->collate_only)) {  // no write-back stages: close the timer's remaining intervals
    for (int i = ST_WRITE_RIGHT; i <= ST_COUNT; ++i) prof_mark(pf, i, st);
    pf->armed = true;
  }
  if (io->embed_only && w.lean) return check_launch("tg_stream_step(embed_only, lean)");  // counts are not written
  if (io->embed_only || io->collate_only) {
    if (io->counts) {
      hipError_t e = hipMemcpyAsync(io->counts, w.counts, 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) {
        set_hip_error(e, "tg_stream_step counts copy");
        return TG_EHIP;
      }
    }
    if (io->offset_dev && io->advance) hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, st, io->offset_dev, io->B);
    return check_launch("tg_stream_step(embed_only)");
  }
  if ((rc = step_writeback_a(m, io, w, st, pf)) != TG_OK) return rc;
  return step_writeback_b(m, g, io, w, st, pf);
}
}  // namespace tg

extern "C" int tg_stream_step(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, void* ws, size_t ws_bytes,
                              void* stream) {
  return tg::stream_step_ext(m, g, io, ws, ws_bytes, tg::as_stream(stream), nullptr, nullptr);
}

// ---------------------------------------------------------------------------------
// Multi-GPU write-back of a global batch (see include/tiger_hip.h)
// ---------------------------------------------------------------------------------
namespace tg {
// batch slice -> positive-node arrays + float32 times, and flag the positive nodes
__global__ void k_wb_prepare(int64_t Bg, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                             const double* __restrict__ ts, const int64_t* __restrict__ eids,
                             const int64_t* __restrict__ off, int64_t* __restrict__ pos, float* __restrict__ ts2f,
                             int64_t* __restrict__ eids_b, uint8_t* __restrict__ flags) {
  const int64_t o = off ? *off : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * Bg; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i < Bg ? i : i - Bg;
    const int64_t node = i < Bg ? src[o + e] : dst[o + e];
    pos[i] = node;
    ts2f[i] = (float)ts[o + e];
    flags[node] = 1;
    if (i < Bg) eids_b[e] = eids[o + e];
  }
}
}  // namespace tg

struct WbWs {
  uint8_t* flags;
  unsigned long long* best;
  int32_t* counts;
  size_t zero_bytes;
  uint64_t* bm;
  uint32_t* rank;
  int64_t *pos, *eids, *uniq, *upos, *index;
  float* ts2f;
  void* scan_ws;
  size_t scan_bytes;
};

static bool carve_wb(const tg_model* m, int64_t Bg, Carver& cv, WbWs& w) {
  const int64_t W = (m->n_nodes + 63) / 64;
  char* z0 = cv.p;
  w.flags = cv.take<uint8_t>((size_t)W * 64);
  w.best = cv.take<unsigned long long>((size_t)2 * Bg);
  w.counts = cv.take<int32_t>(4);
  w.zero_bytes = cv.ok ? (size_t)(cv.p - z0) : 0;
  w.bm = cv.take<uint64_t>((size_t)W);
  w.rank = cv.take<uint32_t>((size_t)W + 1);
  w.pos = cv.take<int64_t>((size_t)2 * Bg);
  w.eids = cv.take<int64_t>((size_t)Bg);
  w.uniq = cv.take<int64_t>((size_t)2 * Bg);
  w.upos = cv.take<int64_t>((size_t)2 * Bg);
  w.index = cv.take<int64_t>((size_t)2 * Bg);
  w.ts2f = cv.take<float>((size_t)2 * Bg);
  w.scan_bytes = tg_unique_compact_workspace_bytes(m->n_nodes);
  w.scan_ws = cv.take<char>(w.scan_bytes);
  return cv.ok;
}

extern "C" size_t tg_stream_writeback_workspace_bytes(const tg_model* m, int64_t Bg) {
  if (!attn_dims_ok(m) || Bg <= 0) return 0;
  const size_t W = (m->n_nodes + 63) / 64, n2 = 2 * (size_t)Bg;
  return align16(W * 64) + align16(n2 * 8) + 16 + align16(W * 8) + align16((W + 1) * 4) + 4 * align16(n2 * 8) +
         align16(Bg * 8) + align16(n2 * 4) + align16(tg_unique_compact_workspace_bytes(m->n_nodes)) + 256;
}

extern "C" int tg_stream_writeback(const tg_model* m, const tg_writeback_io* io, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (!attn_dims_ok(m) || !io || io->Bg <= 0) return TG_EINVAL;
  if (!io->src || !io->dst || !io->ts || !io->eids || !io->rows || !io->left_row || !io->err) return TG_EINVAL;
  if (io->new_from_pending ? !m->pending_vals : !io->new_row) return TG_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t Bg = io->Bg;
  if (m->row_of && (!io->n_upos_dev || !io->owner)) return TG_EUNSUPPORTED;  // rows: the planned, owner-filtered form only
  // planned winners (no dedup work, two launches): the caller hands over the count on the device.  A rank that owns no
  // winner of the batch passes empty lists, whose pointers may be NULL: nothing to write
  if (io->n_upos_dev && !io->upos) return TG_OK;
  if (io->upos) {
    if (!io->index || !io->n_upos_dev || !io->ts32) return TG_EINVAL;
    WritebackArgs wa{};
    wa.B = Bg; wa.src = io->src; wa.dst = io->dst; wa.eids = io->eids; wa.upos = io->upos; wa.index = io->index;
    wa.ts = io->ts32; wa.n_upos = io->n_upos_dev; wa.err = io->err;
    wa.rows = io->rows; wa.new_row = io->new_row; wa.left_row = io->left_row;
    wa.owner = io->owner; wa.my_rank = io->my_rank; wa.new_from_pending = io->new_from_pending;
    int rc;
    if ((rc = writeback_launch(m, wa, 0, st)) != TG_OK) return rc;
    if ((rc = writeback_launch(m, wa, 1, st)) != TG_OK) return rc;
    return check_launch("tg_stream_writeback(planned)");
  }
  Carver cv(ws, ws_bytes);
  WbWs w{};
  if (!carve_wb(m, Bg, cv, w)) return TG_EWORKSPACE;
  hipError_t e = hipMemsetAsync(w.flags, 0, w.zero_bytes, st);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_stream_writeback memset");
    return TG_EHIP;
  }
  hipLaunchKernelGGL(k_wb_prepare, dim3(flat_grid(2 * Bg, 256)), dim3(256), 0, st, Bg, io->src, io->dst, io->ts, io->eids,
                     io->offset_dev, w.pos, w.ts2f, w.eids, w.flags);
  int rc;
  if ((rc = unique_compact_launch(w.flags, w.bm, m->n_nodes, w.rank, w.uniq, w.counts + 0, 2 * Bg, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, w.scan_ws, w.scan_bytes, st)) != TG_OK)
    return rc;
  const int64_t* src = w.pos;
  const int64_t* dst = w.pos + Bg;
  hipLaunchKernelGGL(k_pos_max, dim3(flat_grid(2 * Bg, 256)), dim3(256), 0, st, Bg, src, dst, w.ts2f, w.bm, w.rank, w.best,
                     w.counts + 2);
  hipLaunchKernelGGL(k_pos_winners, dim3(flat_grid(2 * Bg, 256)), dim3(256), 0, st, Bg, src, dst, w.ts2f, w.bm, w.rank,
                     w.best, w.upos, w.index, w.counts + 2);
  WritebackArgs wa{};
  wa.B = Bg; wa.src = src; wa.dst = dst; wa.eids = w.eids; wa.upos = w.upos; wa.index = w.index; wa.ts = w.ts2f;
  wa.n_upos = w.counts + 2; wa.err = io->err;
  wa.rows = io->rows; wa.new_row = io->new_row; wa.left_row = io->left_row; wa.plan_off = io->offset_dev;
  wa.owner = io->owner; wa.my_rank = io->my_rank; wa.new_from_pending = io->new_from_pending;
  if ((rc = writeback_launch(m, wa, 0, st)) != TG_OK) return rc;
  if ((rc = writeback_launch(m, wa, 1, st)) != TG_OK) return rc;
  // phase 1 itself reads the offset (plan_off), so it is advanced by a separate launch
  if (io->offset_dev && io->advance) hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, st, io->offset_dev, Bg);
  return check_launch("tg_stream_writeback");
}
