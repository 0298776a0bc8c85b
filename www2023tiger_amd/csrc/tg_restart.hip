// SeqRestarter forward (tiger/model/restarters.py:51-114).
//
// The reference runs a full self-attention over the last H events and then takes the
// MEAN over the H outputs.  The mean is linear, so only the column means of the
// attention matrix are needed:  mean_t(out_t) = Wo concat_h(Wv_h (sum_s abar_h[s] x_s) + bv_h) + bo
// with abar_h[s] = (1/H) sum_t A_h[t,s]  (rows of A sum to one, so bv passes through).
// Q and K still need every position (H x H scores); V and the output projection collapse
// to one row per node.  Identical maths, ~half the flops.
//
// Round 5: the Q / K projection - 90 % of the restarter's flops - runs on a COMPACT operand.
//  * rows.  The input row of the LAST event of every node is [0 .. 0 | TE(0)] (restarters.py:98-103 zero its first dm - d
//    columns, and its time difference to itself is 0): ONE constant row for the whole batch (row 0).  All padded slots of
//    a node (history shorter than H: id, edge, time, direction 0) hold the same event, hence the same row: ONE row per
//    node.  Everything else is a row of its own.  slot_row[i * H + t] maps the H x H score grid back onto these rows.
//  * columns.  x = [nfeat[src] | nfeat[dst] | anony_emb[anon] | efeat[eid] | TE(dt)].  When the node-feature table is all
//    zeros (every JODIE set: Wikipedia, Reddit, LastFM, MOOC) the first 2d columns meet zeros; the anony_emb block takes
//    H + 1 values only, so its projection is a table T_a = anony_emb W[:, 2d:3d]^T ([H + 1, 2 dm], per call) that the score
//    kernels add when they load a q / k row.  The product keeps K = d_e + d of 5d ("narrow" form).  With a non-zero
//    node-feature table the operand keeps all five blocks ("wide" form) and only the row compaction applies.
// The same sum in another order: float32 reassociation only (the fixtures hold at 1e-4 as before).
#include <algorithm>

#include "tg_sample.h"
#include "tg_step.h"

namespace tg {

__device__ __forceinline__ float4 seq_te4(const float4* __restrict__ fq, const float4* __restrict__ ph, float dt, int cc) {
  const float4 w = fq[cc], q = ph[cc];
  return make_float4(time_enc(dt, w.x, q.x), time_enc(dt, w.y, q.y), time_enc(dt, w.z, q.z), time_enc(dt, w.w, q.w));
}

// compact rows of node i besides the shared constant row: its real events among t < H - 1, plus one row for its padded slots
__global__ void __launch_bounds__(256) k_seq_count(int64_t n, const int32_t* __restrict__ n_dev, int H,
                                                   const int64_t* __restrict__ h_n, int32_t* __restrict__ cnt) {
  const int64_t live = n_dev ? min(n, (int64_t)*n_dev) : n;
  const int lane = lane_id();
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
    int real = 0;
    if (i < live)
      for (int t0 = 0; t0 < H - 1; t0 += TG_WAVE) {
        const int t = t0 + lane;
        real += __popcll(__ballot(t < H - 1 && h_n[i * H + t] != 0));
      }
    if (lane == 0) cnt[i] = i < live ? real + (real < H - 1 ? 1 : 0) : 0;
  }
}

// base[i] = 1 + sum_{j < i} cnt[j] (row 0 is the constant row); rows = {live compact rows, ... without the constant row}
__global__ void __launch_bounds__(1024) k_seq_scan(int64_t n, const int32_t* __restrict__ cnt, int32_t* __restrict__ base,
                                                   int32_t* __restrict__ rows) {
  __shared__ int32_t part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t lo = min(n, (int64_t)tid * per), hi = min(n, lo + per);
  int32_t s = 0;
  for (int64_t j = lo; j < hi; ++j) s += cnt[j];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int32_t v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int32_t run = 1 + (tid ? part[tid - 1] : 0);
  for (int64_t j = lo; j < hi; ++j) {
    base[j] = run;
    run += cnt[j];
  }
  if (tid == 1023) {
    base[n] = 1 + part[1023];
    rows[0] = 1 + part[1023];
    rows[1] = part[1023];
  }
}

// One block per node: the slot -> row map, the primary slot of every row, and the operand rows themselves.
//   narrow: Xc[row] = [efeat[eid] | TE_r(ts_last - ts_t)] (wx = d_e + d) and the one-hot row of the slot's anonymised id
//           (the weight gradient of the tabulated anony_emb block is a product with it);
//   WIDE:   Xc[row] = [nfeat[src] | nfeat[dst] | anony_emb[anon] | efeat[eid] | TE_r] (wx = dm).
template <bool WIDE>
__global__ void __launch_bounds__(256) k_seq_build_c(tg_model m, tg_seq_restarter r, int64_t n,
                                                     const int32_t* __restrict__ n_dev, const int64_t* __restrict__ nids,
                                                     const int64_t* __restrict__ h_n, const int64_t* __restrict__ anon,
                                                     const int64_t* __restrict__ h_e, const float* __restrict__ h_t,
                                                     const int64_t* __restrict__ h_d, const int32_t* __restrict__ base,
                                                     int32_t* __restrict__ slot_row, int32_t* __restrict__ row_slot,
                                                     int32_t* __restrict__ row_anon, float4* __restrict__ Xc,
                                                     float* __restrict__ oh, int ohw, float* __restrict__ prev_ts) {
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int H = r.hist_len, d4 = m.d / 4, e4 = m.d_e / 4;
  const int wx4 = WIDE ? 4 * d4 + e4 : e4 + d4;
  const int lane = lane_id();
  const float4* nf = reinterpret_cast<const float4*>(m.nfeats);
  const float4* ef = reinterpret_cast<const float4*>(m.efeats);
  const float4* ae = reinterpret_cast<const float4*>(r.anony_emb);
  const float4* fq = reinterpret_cast<const float4*>(r.te_freq);
  const float4* ph = reinterpret_cast<const float4*>(r.te_phase);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int wv = threadIdx.x >> 6;
  if (blockIdx.x == 0 && wv == 0) {  // row 0: the last event of every node, [0 .. 0 | TE(0)]
    for (int c = lane; c < wx4; c += TG_WAVE) Xc[c] = c >= wx4 - d4 ? seq_te4(fq, ph, 0.f, c - (wx4 - d4)) : z;
    if (!WIDE)
      for (int c = lane; c < ohw; c += TG_WAVE) oh[c] = 0.f;
    if (lane == 0) {
      row_slot[0] = -1;
      row_anon[0] = -1;  // the last event carries no anonymised id (its first dm - d columns are zeroed)
    }
  }
  // one BLOCK per node: every wavefront derives the map (a few ballots), wavefront 0 stores it, the rows are dealt to the four
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t so = i * H;
    const int b0 = base[i];
    int npad = 0, firstpad = -1;
    for (int t0 = 0; t0 < H - 1; t0 += TG_WAVE) {
      const int t = t0 + lane;
      const unsigned long long pm = __ballot(t < H - 1 && h_n[so + t] == 0);
      if (pm) {
        if (firstpad < 0) firstpad = t0 + __ffsll((long long)pm) - 1;
        npad += __popcll(pm);
      }
    }
    const int start = b0 + (npad > 0 ? 1 : 0);
    const float t_last = h_t[so + H - 1];
    const int64_t self = nids[i];
    int before = 0, deal = 0;
    for (int t0 = 0; t0 < H; t0 += TG_WAVE) {
      const int t = t0 + lane;
      const bool in = t < H - 1;
      const bool nz = in && h_n[so + t] != 0;
      const unsigned long long mask = __ballot(nz);
      if (wv == 0 && t < H) {
        int row = 0;  // the last slot: the constant row
        if (nz) row = start + before + __popcll(mask & ((1ull << lane) - 1ull));
        else if (in) row = b0;
        slot_row[so + t] = row;
        if (nz || (in && t == firstpad)) {
          row_slot[row] = (int32_t)(so + t);
          row_anon[row] = (int32_t)anon[so + t];  // 0: the padded row (utils.py:19-27 numbers real ids from 1)
        }
      }
      // the rows of this chunk's primary slots, one at a time, each written by one wavefront
      unsigned long long prim = mask;
      if (firstpad >= t0 && firstpad < t0 + TG_WAVE) prim |= 1ull << (firstpad - t0);
      while (prim) {
        const int b = __ffsll((long long)prim) - 1;
        prim &= prim - 1ull;
        if ((deal++ & 3) != wv) continue;
        const bool pad = !((mask >> b) & 1ull);
        const int row = pad ? b0 : start + before + __popcll(mask & ((1ull << b) - 1ull));
        const int64_t s = so + t0 + b;
        float4* xr = Xc + (int64_t)row * wx4;
        const float dt = t_last - h_t[s];
        const int64_t eid = h_e[s];
        if (WIDE) {
          const int64_t oth = h_n[s];
          const bool dir = h_d[s] != 0;  // the query node was the destination (graph.py:239-240)
          const int64_t s_n = dir ? self : oth, d_n = dir ? oth : self;
          const int64_t a = anon[s];
          for (int c = lane; c < wx4; c += TG_WAVE) {
            float4 v;
            if (c < d4) v = nf ? nf[s_n * d4 + c] : z;
            else if (c < 2 * d4) v = nf ? nf[d_n * d4 + (c - d4)] : z;
            else if (c < 3 * d4) v = ae[a * d4 + (c - 2 * d4)];
            else if (c < 3 * d4 + e4) v = ef ? ef[eid * e4 + (c - 3 * d4)] : z;
            else v = seq_te4(fq, ph, dt, c - 3 * d4 - e4);
            xr[c] = v;
          }
        } else {
          for (int c = lane; c < wx4; c += TG_WAVE) xr[c] = c < e4 ? (ef ? ef[eid * e4 + c] : z) : seq_te4(fq, ph, dt, c - e4);
          const int a = (int)anon[s];
          for (int c = lane; c < ohw; c += TG_WAVE) oh[(int64_t)row * ohw + c] = c == a ? 1.f : 0.f;
        }
      }
      before += __popcll(mask);
    }
    if (wv == 0 && lane == 0) prev_ts[i] = t_last;
  }
}

// ---- attention scores on the COMPACT rows of a node -------------------------------------------------------------------
// Node i owns lc = cnt[i] + 1 operand rows: [its padded row (if it has padded slots) | its real events | the shared last
// row].  The H x H attention of the reference has only lc x lc distinct entries: the P padded query slots share row 0 (and
// as keys they are masked), so
//   abar[s] = (1 / H) sum_r M[r, s] A[r, s],   M[r, s] = multiplicity of query row r (P for the padded row, else 1)
// and with attention dropout M[r, s] = scale * (number of slots t of row r whose mask entry (t, slot(s)) is kept).
struct SeqRows {
  const float* qk;              // [rows, 2 dm] compact
  const int32_t* slot_row;      // [n * H]
  const int32_t* row_slot;      // [rows] primary slot of a compact row
  const int32_t* row_anon;      // [rows] its anonymised id (0: the node's padded row, -1: the shared last row)
  const int32_t* base;          // [n + 1] first compact row of a node
  const float* ta;              // nullable: [H + 1, 2 dm] tabulated anony_emb part of q / k (not for the last event)
  const int64_t* anon;          // [n * H]
};

typedef float seq_f32x16 __attribute__((ext_vector_type(16)));
constexpr int SQ_CH = 64;  // dh chunk staged per iteration
constexpr int SQ_LD = 66;  // LDS row stride of a staged chunk: the MFMA operand reads of Q K^T (row = lane & 31, column + (lane >> 5)) are conflict-free

template <int HP>
struct SeqLds {
  int64_t s_off[HP], s_toff[HP];
  float sq[HP][SQ_LD], sk[HP][SQ_LD];
  float sc[HP][HP + 2];
  float colm[HP], mult[HP];
  int cslot[HP], padkeep[HP];
  int lc, haspad;
};

// chunk c0 of the q / k rows: global -> registers (seq_fetch), registers -> LDS (seq_commit), so that the loads of the NEXT
// chunk are in flight while the matrix cores work on this one.  8-byte loads (V2: dh even, so every row piece is 8-byte
// aligned), unconditional (clamped row and column; the tabulated part from a valid dummy address where it does not apply).
template <int HP>
struct SeqPf {
  float2 q[HP / 8], k[HP / 8], tq[HP / 8], tk[HP / 8];
};
template <int HP, bool V2>
__device__ __forceinline__ void seq_fetch(const SeqRows& sr, const SeqLds<HP>& L, int lc, int dh, int dm, int c0, int tid,
                                          SeqPf<HP>& p) {
#pragma unroll
  for (int u = 0; u < HP / 8; ++u) {
    if (u * 256 < lc * (SQ_CH / 2)) {  // (block-uniform)
      const int f = tid + u * 256;
      const int row = min(f / (SQ_CH / 2), lc - 1), cc = (f % (SQ_CH / 2)) * 2;
      const int64_t to = L.s_toff[row];
      if (V2) {
        const int col = min(c0 + cc, dh - 2);
        const float* b = sr.qk + L.s_off[row] + col;
        const float* tb = to >= 0 ? sr.ta + to + col : b;
        p.q[u] = *reinterpret_cast<const float2*>(b);
        p.k[u] = *reinterpret_cast<const float2*>(b + dm);
        p.tq[u] = *reinterpret_cast<const float2*>(tb);
        p.tk[u] = *reinterpret_cast<const float2*>(tb + dm);
      } else {
        const int col0 = min(c0 + cc, dh - 1), col1 = min(c0 + cc + 1, dh - 1);
        const float* b = sr.qk + L.s_off[row];
        const float* tb = to >= 0 ? sr.ta + to : b;
        p.q[u] = make_float2(b[col0], b[col1]);
        p.k[u] = make_float2(b[dm + col0], b[dm + col1]);
        p.tq[u] = make_float2(tb[col0], tb[col1]);
        p.tk[u] = make_float2(tb[dm + col0], tb[dm + col1]);
      }
    }
  }
}
template <int HP>
__device__ __forceinline__ void seq_commit(SeqLds<HP>& L, int lc, int dh, int c0, int tid, const SeqPf<HP>& p) {
#pragma unroll
  for (int u = 0; u < HP / 8; ++u) {
    const int f = tid + u * 256;
    const int row = f / (SQ_CH / 2), cc = (f % (SQ_CH / 2)) * 2;
    if (row < lc) {
      const float tm = L.s_toff[row] >= 0 ? 1.f : 0.f;
      const bool ok0 = c0 + cc < dh, ok1 = c0 + cc + 1 < dh;
      L.sq[row][cc] = ok0 ? fmaf(tm, p.tq[u].x, p.q[u].x) : 0.f;
      L.sq[row][cc + 1] = ok1 ? fmaf(tm, p.tq[u].y, p.q[u].y) : 0.f;
      L.sk[row][cc] = ok0 ? fmaf(tm, p.tk[u].x, p.k[u].x) : 0.f;
      L.sk[row][cc + 1] = ok1 ? fmaf(tm, p.tk[u].y, p.k[u].y) : 0.f;
    }
  }
}

// the node's rows, their slots and multiplicities
template <int HP>
__device__ __forceinline__ void seq_lds_init(const SeqRows& sr, SeqLds<HP>& L, const int64_t* __restrict__ h_n, int64_t i,
                                             int h, int H, int dh, int dm, int tid) {
  for (int f = tid; f < HP * SQ_LD; f += 256) {  // rows past lc stay zero: they are multiplied by zeros of the grid
    (&L.sq[0][0])[f] = 0.f;
    (&L.sk[0][0])[f] = 0.f;
  }
  for (int f = tid; f < HP * (HP + 2); f += 256) (&L.sc[0][0])[f] = 0.f;
  const int b0 = sr.base[i], lc = sr.base[i + 1] - b0 + 1;
  // row 0 is the padded row iff the node has padded slots (its anonymised id is 0; real events count from 1)
  const int haspad = (lc > 1 && sr.row_anon[b0] == 0) ? 1 : 0;
  if (tid == 0) {
    L.lc = lc;
    L.haspad = haspad;
  }
  if (tid < lc) {
    const int grow = tid < lc - 1 ? b0 + tid : 0;
    const int an = sr.row_anon[grow];
    const int slot = tid < lc - 1 ? sr.row_slot[grow] - (int32_t)(i * H) : H - 1;
    L.s_off[tid] = (int64_t)grow * 2 * dm + (int64_t)h * dh;
    L.s_toff[tid] = (sr.ta && an >= 0) ? (int64_t)an * 2 * dm + (int64_t)h * dh : -1;
    L.cslot[tid] = slot;
    L.mult[tid] = (tid == 0 && haspad) ? (float)(H - 1 - (lc - 2)) : 1.f;  // padded slots = (H - 1) - real events
    L.padkeep[tid] = 0;
  }
  __syncthreads();
}

// S = Q K^T of the node's rows on v_mfma_f32_32x32x2_f32 (lane l feeds A[l & 31][l >> 5], B[l >> 5][l & 31]), scaled, the
// padded key (restarters.py:86-87) at -inf, left in L.sc.  HP = 64: every wavefront takes a quarter of each chunk's k-steps
// for ALL tiles (one tile for most nodes: four wavefronts share its serial MFMA chain), partial grids added in wavefront
// order; HP = 128: a wavefront owns a row of tiles.
template <int HP, bool V2>
__device__ __forceinline__ void seq_scores_to_lds(const SeqRows& sr, SeqLds<HP>& L, int dh, int dm, int tid) {
  constexpr int NTM = HP / 32;
  constexpr bool KS = HP <= 64;
  constexpr int NA = KS ? NTM * NTM : NTM;
  const int lc = L.lc, NT = (lc + 31) / 32;
  const int wave = tid >> 6, lane = tid & 63, fr = lane & 31, fk = lane >> 5;
  seq_f32x16 acc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  SeqPf<HP> pf;
  seq_fetch<HP, V2>(sr, L, lc, dh, dm, 0, tid, pf);
  for (int c0 = 0; c0 < dh; c0 += SQ_CH) {
    seq_commit<HP>(L, lc, dh, c0, tid, pf);
    __syncthreads();
    if (c0 + SQ_CH < dh) seq_fetch<HP, V2>(sr, L, lc, dh, dm, c0 + SQ_CH, tid, pf);
    if (KS) {
#pragma unroll
      for (int kq = 0; kq < SQ_CH / 8; ++kq) {
        const int kk = 2 * (wave + 4 * kq);
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
          if (tm < NT) {
            const float a = L.sq[tm * 32 + fr][kk + fk];
#pragma unroll
            for (int tn = 0; tn < NTM; ++tn)
              if (tn < NT)
                acc[(KS ? tm * NTM : 0) + tn] =
                    __builtin_amdgcn_mfma_f32_32x32x2f32(a, L.sk[tn * 32 + fr][kk + fk], acc[(KS ? tm * NTM : 0) + tn], 0, 0, 0);
          }
      }
    } else if (wave < NT) {
#pragma unroll 4
      for (int kk = 0; kk < SQ_CH; kk += 2) {
        const float a = L.sq[wave * 32 + fr][kk + fk];
#pragma unroll
        for (int tn = 0; tn < NTM; ++tn)
          if (tn < NT) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, L.sk[tn * 32 + fr][kk + fk], acc[tn], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const float scale = 1.0f / sqrtf((float)dh);
  if (KS) {
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
          for (int tn = 0; tn < NTM; ++tn)
            if (tm < NT && tn < NT) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int t = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, sidx = tn * 32 + fr;
                if (t < lc && sidx < lc) L.sc[t][sidx] += acc[(KS ? tm * NTM : 0) + tn][r];
              }
            }
      }
      __syncthreads();
    }
    for (int p = tid; p < lc * lc; p += 256) {
      const int t = p / lc, sidx = p % lc;
      L.sc[t][sidx] = (sidx == 0 && L.haspad) ? -INFINITY : L.sc[t][sidx] * scale;
    }
  } else {
    if (wave < NT) {
#pragma unroll
      for (int tn = 0; tn < NTM; ++tn)
        if (tn < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, sidx = tn * 32 + fr;
            if (t < lc && sidx < lc) L.sc[t][sidx] = (sidx == 0 && L.haspad) ? -INFINITY : acc[tn][r] * scale;
          }
        }
    }
  }
  __syncthreads();
}

// attention dropout (nn.MultiheadAttention): padkeep[s] = number of padded slots t whose mask entry (t, slot(s)) is kept
template <int HP>
__device__ __forceinline__ void seq_padkeep(SeqLds<HP>& L, const int64_t* __restrict__ h_n, int64_t i, int h, int H, int nh,
                                            uint64_t dkey, const DropCfg& dc, int tid) {
  if (dc.p > 0.f && L.haspad) {
    const int lc = L.lc;
    for (int p = tid; p < (H - 1) * lc; p += 256) {
      const int t = p / lc, sidx = p % lc;
      if (h_n[i * H + t] == 0 &&
          drop_keep(dkey, DROP_SEQ_ATTN, (((uint64_t)i * nh + h) * H + t) * H + L.cslot[sidx], dc.thresh))
        atomicAdd(&L.padkeep[sidx], 1);
    }
  }
  __syncthreads();
}
// M[r, s] (see above)
template <int HP>
__device__ __forceinline__ float seq_mult(const SeqLds<HP>& L, int r, int sidx, int64_t i, int h, int H, int nh, uint64_t dkey,
                                          const DropCfg& dc) {
  if (dc.p <= 0.f) return L.mult[r];
  if (r == 0 && L.haspad) return dc.scale * (float)L.padkeep[sidx];
  return drop_keep(dkey, DROP_SEQ_ATTN, (((uint64_t)i * nh + h) * H + L.cslot[r]) * H + L.cslot[sidx], dc.thresh) ? dc.scale : 0.f;
}

// One block per (node, head): scores, key padding mask, row softmax, column mean - on the node's own rows.
template <int HP, bool V2>
__global__ void __launch_bounds__(256) k_seq_scores(int64_t n, int H, int dm, int nh, SeqRows sr,
                                                    const int64_t* __restrict__ h_n, float* __restrict__ abar,
                                                    const int32_t* __restrict__ n_dev, DropCfg dc,
                                                    float* __restrict__ rbar, int lc_min) {
  if (n_dev && (int64_t)(blockIdx.x / nh) >= (int64_t)*n_dev) return;
  {  // this launch serves the nodes with lc_min < rows <= HP (small grids: less LDS, more blocks per CU)
    const int lc0 = sr.base[blockIdx.x / nh + 1] - sr.base[blockIdx.x / nh] + 1;
    if (lc0 <= lc_min || lc0 > HP) return;
  }
  extern __shared__ __align__(16) unsigned char seq_lds_raw[];
  SeqLds<HP>& L = *reinterpret_cast<SeqLds<HP>*>(seq_lds_raw);
  const uint64_t dkey = drop_key(dc);
  const int64_t i = blockIdx.x / nh;
  const int h = blockIdx.x % nh;
  const int dh = dm / nh;
  const int tid = threadIdx.x;
  seq_lds_init<HP>(sr, L, h_n, i, h, H, dh, dm, tid);
  seq_scores_to_lds<HP, V2>(sr, L, dh, dm, tid);
  seq_padkeep<HP>(L, h_n, i, h, H, nh, dkey, dc, tid);
  const int lc = L.lc;
  if (tid < lc) {  // row softmax, weighted by the row's multiplicity (and dropout mask)
    float mx = -INFINITY;
    for (int s = 0; s < lc; ++s) mx = fmaxf(mx, L.sc[tid][s]);
    float sum = 0.f;
    for (int s = 0; s < lc; ++s) {
      const float e = expf(L.sc[tid][s] - mx);
      L.sc[tid][s] = e;
      sum += e;
    }
    const float inv = 1.f / sum;
    for (int s = 0; s < lc; ++s) L.sc[tid][s] *= inv * seq_mult<HP>(L, tid, s, i, h, H, nh, dkey, dc);
  }
  __syncthreads();
  if (tid < lc) {  // column mean over the H query slots
    float a = 0.f;
    for (int t = 0; t < lc; ++t) a += L.sc[t][tid];
    L.colm[tid] = a / (float)H;
  }
  __syncthreads();
  if (tid < H) {  // back to slot space: a padded slot reads the padded row's column, which is masked (0)
    const int r = tid == H - 1 ? lc - 1 : sr.slot_row[i * H + tid] - sr.base[i];
    abar[((int64_t)i * nh + h) * H + tid] = L.colm[r];
  }
  if (tid == 0 && rbar) {  // sum of the column means: weight of the value bias (1 without dropout)
    float r = 0.f;
    for (int s = 0; s < lc; ++s) r += L.colm[s];
    rbar[(int64_t)i * nh + h] = dc.p > 0.f ? r : 1.f;
  }
}

// where the columns of the full input row x(i, s) live (see the file header)
struct SeqCols {
  const float4* xc;         // compact operand rows
  int wx4, col0_4;          // their width and first column (float4 units): narrow 3d, wide 0
  const float4* ae;         // narrow: anony_emb, columns [2d, 3d) of every slot but the last; wide: nullptr (inside xc)
  const int32_t *base, *row_slot, *row_anon;  // the node's rows, their primary slots and anonymised ids
};

// xbar[i, h, :] = sum_s abar[i, h, s] * x(i, s): over the node's COMPACT rows (a padded slot's weight is 0: its key is
// masked) and the shared last row
__global__ void k_seq_mix(int64_t n, int H, int row4, int d4, int nh, const float* __restrict__ abar, SeqCols sc,
                          float4* __restrict__ xbar, const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int64_t total = n * nh * row4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % row4);
    const int64_t ih = t / row4;
    const int64_t i = ih / nh;
    const int b0 = sc.base[i], b1 = sc.base[i + 1];
    const float* ab = abar + ih * H - i * H;  // indexed by the global slot number
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c >= sc.col0_4 && c < sc.col0_4 + sc.wx4) {
      const int cc = c - sc.col0_4;
      for (int r = b0; r < b1; ++r) {
        const float w = ab[sc.row_slot[r]];
        const float4 x = sc.xc[(int64_t)r * sc.wx4 + cc];
        a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w;
      }
      const float w = abar[ih * H + H - 1];
      const float4 x = sc.xc[cc];
      a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w;
    } else if (sc.ae && c >= 2 * d4 && c < 3 * d4) {
      for (int r = b0; r < b1; ++r) {
        const float w = ab[sc.row_slot[r]];
        const float4 x = sc.ae[(int64_t)sc.row_anon[r] * d4 + (c - 2 * d4)];
        a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w;
      }
    }
    xbar[t] = a;
  }
}

// ---------------------------------------------------------------------------------
// backward kernels (mutual loss, tiger.py:576-590)
// ---------------------------------------------------------------------------------
// dabar[i, h, s] = dxbar[i, h, :] . x(i, s)   (one wavefront per (compact row or last slot, head): the score backward
// reads the entries of the rows' primary slots only)
__global__ void __launch_bounds__(256) k_seq_mix_bwd(int64_t n, const int32_t* __restrict__ n_dev,
                                                     const int32_t* __restrict__ rows_dev, int H, int row4, int d4,
                                                     int nh, const float4* __restrict__ dxbar, SeqCols sc,
                                                     float* __restrict__ dabar, const float* __restrict__ dO,
                                                     const float* __restrict__ bv) {
  // dO / bv non-null (dropout): abar also weights the value bias, d rbar = dO_h . bv_h is added
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int lane = lane_id();
  const int64_t nrow = (int64_t)rows_dev[0] - 1;  // compact rows besides the shared last row
  const int64_t total = (nrow + n) * nh;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += (int64_t)gridDim.x * 4) {
    int64_t i, g;
    int sidx;
    const int h = (int)(t % nh);
    if (t < nrow * nh) {
      g = 1 + t / nh;
      const int32_t slot = sc.row_slot[g];
      i = slot / H;
      sidx = slot - (int32_t)(i * H);
    } else {
      g = 0;
      i = (t - nrow * nh) / nh;
      sidx = H - 1;
    }
    const int64_t ih = i * nh + h;
    const float4* a = dxbar + ih * row4;
    const float4* b = sc.xc + g * sc.wx4;
    float acc = 0.f;
    for (int c = lane; c < sc.wx4; c += TG_WAVE) {
      const float4 u = a[sc.col0_4 + c], v = b[c];
      acc = fmaf(u.x, v.x, fmaf(u.y, v.y, fmaf(u.z, v.z, fmaf(u.w, v.w, acc))));
    }
    if (sc.ae && sidx != H - 1) {
      const float4* e = sc.ae + (int64_t)sc.row_anon[g] * d4;
      for (int c = lane; c < d4; c += TG_WAVE) {
        const float4 u = a[2 * d4 + c], v = e[c];
        acc = fmaf(u.x, v.x, fmaf(u.y, v.y, fmaf(u.z, v.z, fmaf(u.w, v.w, acc))));
      }
    }
    if (dO) {
      const int dm = row4 * 4, dh = dm / nh;
      for (int c = lane; c < dh; c += TG_WAVE) acc = fmaf(dO[i * dm + h * dh + c], bv[h * dh + c], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) dabar[ih * H + sidx] = acc;
  }
}

// One block per (node, head): recompute the attention of the node's rows (as k_seq_scores), then
//   dA[r, s] = M[r, s] dabar[s] / H;  dS[r, s] = A[r, s] (dA[r, s] - sum_s' A[r, s'] dA[r, s'])
//   dq_r = scale sum_s dS[r, s] k_s;   dk_s = scale sum_r dS[r, s] q_r      (both on the matrix cores, per dh chunk)
// The gradients land on the COMPACT rows directly (the padded row's is the sum over its slots: M carries the multiplicity;
// its key is masked: zero key gradient), the last row's in dqk_last [n, 2 dm] (summed over the nodes into row 0 afterwards).
template <int HP, bool V2>
__global__ void __launch_bounds__(256) k_seq_scores_bwd(int64_t n, const int32_t* __restrict__ n_dev, int H, int dm,
                                                        int nh, SeqRows sr, const int64_t* __restrict__ h_n,
                                                        const float* __restrict__ dabar, float* __restrict__ dqk,
                                                        float* __restrict__ dqk_last, DropCfg dc, int lc_min) {
  if (n_dev && (int64_t)(blockIdx.x / nh) >= (int64_t)*n_dev) return;
  {
    const int lc0 = sr.base[blockIdx.x / nh + 1] - sr.base[blockIdx.x / nh] + 1;
    if (lc0 <= lc_min || lc0 > HP) return;
  }
  extern __shared__ __align__(16) unsigned char seq_lds_raw[];
  SeqLds<HP>& L = *reinterpret_cast<SeqLds<HP>*>(seq_lds_raw);
  const uint64_t dkey = drop_key(dc);
  const int64_t i = blockIdx.x / nh;
  const int h = blockIdx.x % nh;
  const int dh = dm / nh;
  const int tid = threadIdx.x;
  seq_lds_init<HP>(sr, L, h_n, i, h, H, dh, dm, tid);
  seq_scores_to_lds<HP, V2>(sr, L, dh, dm, tid);
  seq_padkeep<HP>(L, h_n, i, h, H, nh, dkey, dc, tid);
  const int lc = L.lc;
  const float scale = 1.0f / sqrtf((float)dh);
  if (tid < lc) {  // row softmax, then the softmax backward of that row, pre-multiplied by the score scale
    float mx = -INFINITY;
    for (int s = 0; s < lc; ++s) mx = fmaxf(mx, L.sc[tid][s]);
    float sum = 0.f;
    for (int s = 0; s < lc; ++s) {
      const float e = expf(L.sc[tid][s] - mx);
      L.sc[tid][s] = e;
      sum += e;
    }
    const float inv = 1.f / sum, invH = 1.f / (float)H;
    const float* da = dabar + ((int64_t)i * nh + h) * H;
    auto dA = [&](int s) { return da[L.cslot[s]] * invH * seq_mult<HP>(L, tid, s, i, h, H, nh, dkey, dc); };
    float dot = 0.f;
    for (int s = 0; s < lc; ++s) {
      const float a = L.sc[tid][s] * inv;
      L.sc[tid][s] = a;
      dot = fmaf(a, dA(s), dot);
    }
    for (int s = 0; s < lc; ++s) L.sc[tid][s] = L.sc[tid][s] * (dA(s) - dot) * scale;
  }
  __syncthreads();
  // per chunk of 64 columns: a wavefront takes dq or dk (wave & 1) of one 32-column half (wave >> 1), row tile by row tile:
  // dq tile = dS[rows, :] K[:, half],  dk tile = dS[:, rows]^T Q[:, half]; the contraction runs over the lc rows
  const int wave = tid >> 6, lane = tid & 63, fr = lane & 31, fk = lane >> 5;
  const int isk = wave & 1, half = wave >> 1;
  const int NT = (lc + 31) / 32, Lk = (lc + 1) & ~1;
  SeqPf<HP> pf;
  seq_fetch<HP, V2>(sr, L, lc, dh, dm, 0, tid, pf);
  for (int c0 = 0; c0 < dh; c0 += SQ_CH) {
    seq_commit<HP>(L, lc, dh, c0, tid, pf);
    __syncthreads();
    if (c0 + SQ_CH < dh) seq_fetch<HP, V2>(sr, L, lc, dh, dm, c0 + SQ_CH, tid, pf);
    const int col = c0 + half * 32 + fr;
    for (int tm = 0; tm < NT; ++tm) {
      const int r0 = tm * 32;
      seq_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if (!isk) {
        for (int kk = 0; kk < Lk; kk += 2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(L.sc[r0 + fr][kk + fk], L.sk[kk + fk][half * 32 + fr], acc, 0, 0, 0);
      } else {
        for (int kk = 0; kk < Lk; kk += 2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(L.sc[kk + fk][r0 + fr], L.sq[kk + fk][half * 32 + fr], acc, 0, 0, 0);
      }
      if (col < dh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * fk;
          if (row < lc) {
            float* ob = (row == lc - 1 ? dqk_last + i * 2 * dm + (int64_t)h * dh : dqk + L.s_off[row]) + col;
            ob[isk ? dm : 0] = acc[r];
          }
        }
      }
    }
    __syncthreads();
  }
}

// gradient of the input rows that carry parameters: the anonymised-position embedding
// (columns [2d, 3d), not for the zeroed last event) and the restarter's TimeEncode (last d
// columns).  dX = dXs (from the q/k projection, per COMPACT row: taken by the row's primary slot - the constant row's by
// the last slot of node 0) + sum_h abar_h dxbar_h (from the value mix, per slot).  Narrow form: the q/k projection's
// anony_emb gradient goes through the table T_a (k_seq_anon_grads), dXs holds the TimeEncode block only (ldx = d);
// wide form: dXs = [anony_emb block | TimeEncode block] (ldx = 2d).
__global__ void __launch_bounds__(256) k_seq_build_bwd(tg_model m, tg_seq_restarter r, int64_t n,
                                                       const int32_t* __restrict__ n_dev,
                                                       const int64_t* __restrict__ anon, const float* __restrict__ h_t,
                                                       const int32_t* __restrict__ slot_row,
                                                       const int32_t* __restrict__ row_slot,
                                                       const float* __restrict__ dXs, int wide,
                                                       const float* __restrict__ abar,
                                                       const float* __restrict__ dxbar, int use_lds,
                                                       float* __restrict__ danon, float* __restrict__ bpart) {
  // A thread OWNS column c = threadIdx.x (+ 256, ...) and walks the block's slots, two at a time: the TimeEncode sums stay
  // in registers and the embedding rows in LDS, one writer per element - no atomics.  The block leaves its sums in
  // bpart[block] = [d freq | d phase | (H + 1) x d embedding rows]; k_seq_build_reduce adds the blocks in order.
  extern __shared__ float lan[];  // [(H + 1), d] embedding grads when use_lds
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int H = r.hist_len, d = m.d, dm = 4 * d + m.d_e, nh = r.n_head;
  const int ldx = wide ? 2 * d : d, tcol = wide ? d : 0;
  const int nl = use_lds ? (H + 1) * d : 0;
  for (int c = threadIdx.x; c < nl; c += 256) lan[c] = 0.f;
  __syncthreads();
  const int64_t slots = n * H;
  const int64_t per = (slots + gridDim.x - 1) / gridDim.x;
  const int64_t lo = min(slots, (int64_t)blockIdx.x * per), hi = min(slots, lo + per);
  float* bp = bpart + (int64_t)blockIdx.x * (2 * d + nl);
  for (int c = threadIdx.x; c < d; c += 256) {
    const float fw = r.te_freq[c], fp = r.te_phase[c];
    float sf = 0.f, sp = 0.f;
    auto grads = [&](int64_t row, float& ga, float& gt, float& dt, int64_t& a) {
      const int64_t i = row / H;
      const int pos = (int)(row - i * H);
      const int32_t cr = slot_row[row];
      const bool primary = pos == H - 1 ? i == 0 : row_slot[cr] == (int32_t)row;
      const float xt = dXs[(int64_t)cr * ldx + tcol + c], xa = dXs[(int64_t)cr * ldx + c];  // (unconditional loads)
      gt = primary ? xt : 0.f;
      ga = (primary && wide) ? xa : 0.f;
      for (int h = 0; h < nh; ++h) {
        const float w = abar[((int64_t)i * nh + h) * H + pos];
        const float* dx = dxbar + ((int64_t)i * nh + h) * dm;
        ga = fmaf(w, dx[2 * d + c], ga);
        gt = fmaf(w, dx[3 * d + m.d_e + c], gt);
      }
      a = pos != H - 1 ? anon[row] : -1;
      dt = h_t[i * H + H - 1] - h_t[row];
    };
    auto apply = [&](float ga, float gt, float dt, int64_t a) {
      if (a >= 0) {
        if (use_lds) lan[a * d + c] += ga;
        else atomicAdd(danon + a * d + c, ga);
      }
      const float sn = -time_enc_sin(dt, fw, fp) * gt;
      sf = fmaf(sn, dt, sf);
      sp += sn;
    };
    int64_t row = lo;
    for (; row + 1 < hi; row += 2) {
      float ga0, gt0, dt0, ga1, gt1, dt1;
      int64_t a0, a1;
      grads(row, ga0, gt0, dt0, a0);
      grads(row + 1, ga1, gt1, dt1, a1);
      apply(ga0, gt0, dt0, a0);
      apply(ga1, gt1, dt1, a1);
    }
    if (row < hi) {
      float ga0, gt0, dt0;
      int64_t a0;
      grads(row, ga0, gt0, dt0, a0);
      apply(ga0, gt0, dt0, a0);
    }
    bp[c] = sf;
    bp[d + c] = sp;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < nl; c += 256) bp[2 * d + c] = lan[c];
}
// dfreq / dphase / danon += the blocks' sums in a fixed order: 64 elements per workgroup, its 16 wavefronts each add a
// sixteenth of the blocks (lanes = consecutive elements), wavefront sums folded in wavefront order
__global__ void __launch_bounds__(1024) k_seq_build_reduce(int nb, int d, int nl, const float* __restrict__ bpart,
                                                           float* __restrict__ dfreq, float* __restrict__ dphase,
                                                           float* __restrict__ danon) {
  __shared__ float red[16][64];
  const int w = 2 * d + nl, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const int per = (nb + 15) / 16, b0 = min(nb, wv * per), b1 = min(nb, b0 + per);
  float a = 0.f;
  if (e < w)
    for (int b = b0; b < b1; ++b) a += bpart[(int64_t)b * w + e];
  red[wv][lane] = a;
  __syncthreads();
  if (wv == 0 && e < w) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][lane];
    if (e < d) dfreq[e] += t;
    else if (e < 2 * d) dphase[e - d] += t;
    else danon[e - 2 * d] += t;
  }
}

// mutual loss (tiger.py:582-590): MSE over the rows of cat[sur_left, sur_right] whose target
// row is not all zero.  Pass A: row validity, squared error, counts.  Pass B: loss and d pred.
__global__ void __launch_bounds__(256) k_mutual_a(int64_t cap, const int32_t* __restrict__ n_dev, int d,
                                                  const int64_t* __restrict__ index, const float* __restrict__ hpl,
                                                  const float* __restrict__ hpr, const float* __restrict__ sl,
                                                  const float* __restrict__ sr, uint8_t* __restrict__ valid,
                                                  float* __restrict__ acc /* [2]: sq-err sum, valid rows */) {
  const int64_t n = min(cap, (int64_t)*n_dev);
  const int lane = lane_id();
  float se = 0.f, nv = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < 2 * n; t += (int64_t)gridDim.x * 4) {
    const bool right = t >= n;
    const int64_t i = right ? t - n : t;
    const float* tg = (right ? hpr : hpl) + index[i] * d;
    const float* pr = (right ? sr : sl) + i * d;
    float e = 0.f;
    bool nz = false;
    for (int c = lane; c < d; c += TG_WAVE) {
      const float tv = tg[c], df = pr[c] - tv;
      nz |= tv != 0.f;
      e = fmaf(df, df, e);
    }
    const bool ok = __ballot(nz) != 0ull;
    e = wave_sum(e);
    if (lane == 0) valid[t] = ok;
    if (ok) {
      se += e;
      nv += 1.f;
    }
  }
  // one pair of atomics per workgroup (two per wavefront on one address serialised 2 048 of them: 46 us)
  __shared__ float red[2][4];
  if (lane == 0) {
    red[0][threadIdx.x >> 6] = se;
    red[1][threadIdx.x >> 6] = nv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = red[0][0] + red[0][1] + red[0][2] + red[0][3], c = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (c > 0.f) {
      atomicAdd(acc + 0, s);
      atomicAdd(acc + 1, c);
    }
  }
}
__global__ void __launch_bounds__(256) k_mutual_b(int64_t cap, const int32_t* __restrict__ n_dev, int d,
                                                  const int64_t* __restrict__ index, const float* __restrict__ hpl,
                                                  const float* __restrict__ hpr, const float* __restrict__ sl,
                                                  const float* __restrict__ sr, const uint8_t* __restrict__ valid,
                                                  const float* __restrict__ acc, float* __restrict__ dsl,
                                                  float* __restrict__ dsr, float* __restrict__ loss_out,
                                                  int32_t* __restrict__ flag_out) {
  const int64_t n = min(cap, (int64_t)*n_dev);
  const float nv = acc[1];
  const float inv = nv > 0.f ? 1.f / (nv * (float)d) : 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *loss_out = acc[0] * inv;
    if (flag_out) *flag_out = nv > 0.f;
  }
  const int64_t total = 2 * n * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = e / d;
    const int c = (int)(e - t * d);
    const bool right = t >= n;
    const int64_t i = right ? t - n : t;
    float g = 0.f;
    if (valid[t]) g = 2.f * ((right ? sr : sl)[i * d + c] - (right ? hpr : hpl)[index[i] * d + c]) * inv;
    (right ? dsr : dsl)[i * d + c] = g;
  }
}

// MergeLayer dropout on the hidden activations, in place (basic_modules.py:18)
__global__ void k_dropout_rows(int64_t n, const int32_t* __restrict__ n_dev, int d, float* __restrict__ x, DropCfg dc,
                               uint32_t stream) {
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const uint64_t dkey = drop_key(dc);
  const int64_t total = n * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    x[e] = drop_keep(dkey, stream, (uint64_t)e, dc.thresh) ? x[e] * dc.scale : 0.f;
}

// dOm[h] = dO with the columns outside head h's slice zeroed
__global__ void k_head_mask(int64_t n, const int32_t* __restrict__ n_dev, int dm, int nh, const float* __restrict__ dO,
                            float* __restrict__ dOm) {
  const int64_t cap = n;
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int dh = dm / nh;
  const int64_t total = n * dm;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % dm);
    const float v = dO[e];
    for (int h = 0; h < nh; ++h) dOm[(int64_t)h * cap * dm + e] = (c / dh == h) ? v : 0.f;
  }
}

// StaticRestarter (restarters.py:254-277): surrogate rows are embedding rows of the unique nodes
__global__ void k_static_rows(int64_t cap, const int32_t* __restrict__ n_dev, int d, const int64_t* __restrict__ nids,
                              const float* __restrict__ left, const float* __restrict__ right, float* __restrict__ sl,
                              float* __restrict__ sr) {
  const int64_t total = min(cap, (int64_t)*n_dev) * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / d;
    const int64_t o = nids[i] * d + (e - i * d);
    sl[e] = left[o];
    sr[e] = right[o];
  }
}
__global__ void k_static_rows_bwd(int64_t cap, const int32_t* __restrict__ n_dev, int d,
                                  const int64_t* __restrict__ nids, const float* __restrict__ dsl,
                                  const float* __restrict__ dsr, float* __restrict__ gleft, float* __restrict__ gright) {
  const int64_t total = min(cap, (int64_t)*n_dev) * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / d;
    const int64_t o = nids[i] * d + (e - i * d);  // unique nodes: no collisions
    gleft[o] += dsl[e];
    gright[o] += dsr[e];
  }
}

// restart-data collation on device (data_loader.py:133-142): query arrays for the history
// sampler, padded with node 0 (empty history) past the live count; counts2 = {n, n * H}
__global__ void k_restart_queries(int64_t B, const double* __restrict__ ts, const int64_t* __restrict__ off,
                                  double* __restrict__ ts2) {
  const int64_t o = off ? *off : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * B; i += (int64_t)gridDim.x * blockDim.x)
    ts2[i] = ts[o + (i < B ? i : i - B)];
}
__global__ void k_restart_pad(int64_t cap, const int32_t* __restrict__ n_dev, int H, int64_t* __restrict__ uniq,
                              const int64_t* __restrict__ index, const double* __restrict__ ts2,
                              double* __restrict__ tu, int32_t* __restrict__ counts2, float* __restrict__ acc4) {
  const int64_t n = min(cap, (int64_t)*n_dev);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counts2[0] = (int32_t)n;
    counts2[1] = (int32_t)(n * H);
  }
  if (acc4 && blockIdx.x == 0 && threadIdx.x < 4) acc4[threadIdx.x] = 0.f;  // the loss accumulators of the mutual step
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n) {
      tu[i] = ts2[index[i]];
    } else {
      uniq[i] = 0;
      tu[i] = 0.0;
    }
  }
}

struct SeqWs {
  float *xc, *qk, *abar, *xbar, *o, *om, *t2, *rbar, *ta, *oh;
  int32_t *cnt, *base, *slot_row, *row_slot, *row_anon, *rows;  // the compact-row plan (see the file header); rows = {live rows, ...}
  int64_t rowcap;
  int wx, ohw;
  bool wide;
};

// wide form (all five column blocks in the compact operand): a node-feature table that is not known to be all zeros
static bool seq_wide(const tg_model* m, const tg_seq_restarter* r) { return m->nfeats && !r->nfeats_zero; }

static bool carve_seq(const tg_model* m, const tg_seq_restarter* r, int64_t n, Carver& cv, SeqWs& w, bool keep_t2) {
  const size_t dm = 4 * (size_t)m->d + m->d_e, H = r->hist_len, nh = r->n_head;
  w.wide = seq_wide(m, r);
  w.rowcap = n * (int64_t)H + 1;
  w.wx = w.wide ? (int)dm : m->d_e + m->d;
  w.ohw = (int)((H + 1 + 3) & ~(size_t)3);
  w.xc = cv.take<float>((size_t)w.rowcap * w.wx);
  w.qk = cv.take<float>((size_t)w.rowcap * 2 * dm);
  w.abar = cv.take<float>(n * nh * H);
  w.xbar = cv.take<float>(n * nh * dm);
  w.o = cv.take<float>(n * dm);
  w.om = cv.take<float>(n * dm);
  w.t2 = keep_t2 ? cv.take<float>(n * (size_t)m->d) : w.om;
  w.rbar = keep_t2 ? cv.take<float>(n * nh) : nullptr;
  w.ta = w.wide ? nullptr : cv.take<float>((H + 1) * 2 * dm);
  w.oh = w.wide ? nullptr : cv.take<float>((size_t)w.rowcap * w.ohw);
  w.cnt = cv.take<int32_t>(n);
  w.base = cv.take<int32_t>(n + 1);
  w.slot_row = cv.take<int32_t>(n * H);
  w.row_slot = cv.take<int32_t>((size_t)w.rowcap);
  w.row_anon = cv.take<int32_t>((size_t)w.rowcap);
  w.rows = cv.take<int32_t>(4);
  return cv.ok;
}

// bytes a carve takes: a dry run over an address range that is never touched
template <typename F>
static size_t carve_bytes(F&& carve) {
  Carver cv(reinterpret_cast<void*>((uintptr_t)256), ~(size_t)0 >> 1);
  carve(cv);
  return (~(size_t)0 >> 1) - cv.left;
}

static int seq_ok(const tg_model* m, const tg_seq_restarter* r) {
  if (!m || !r || m->d <= 0 || (m->d % 4) || m->d_e <= 0 || (m->d_e % 4)) return 0;
  const int dm = 4 * m->d + m->d_e;
  // head offsets only index scalar loads / row-aligned weight blocks, so dh need not be a multiple of 4
  if (r->hist_len <= 0 || r->n_head <= 0 || dm % r->n_head) return 0;
  return 1;
}

// the score kernels' dynamic LDS exceeds the 64 KB default of a launch for histories beyond 64 events
static int seq_lds_attr() {
  static int done = 0;  // (per process; the attribute belongs to the function, whichever device runs it)
  if (done) return TG_OK;
  hipError_t e = hipSuccess;
  const void* fns[4] = {reinterpret_cast<const void*>(&k_seq_scores<128, true>), reinterpret_cast<const void*>(&k_seq_scores<128, false>),
                        reinterpret_cast<const void*>(&k_seq_scores_bwd<128, true>), reinterpret_cast<const void*>(&k_seq_scores_bwd<128, false>)};
  for (int j = 0; j < 4 && e == hipSuccess; ++j)
    e = hipFuncSetAttribute(fns[j], hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SeqLds<128>));
  if (e != hipSuccess) {
    set_hip_error(e, "seq_lds_attr");
    return TG_EHIP;
  }
  done = 1;
  return TG_OK;
}

// forward on `cap` rows of which the first *n_dev (nullable: all) are live
static int seq_forward(const tg_model* m, const tg_seq_restarter* r, int64_t n, const int32_t* n_dev,
                       const int64_t* nids, const int64_t* h_n, const int64_t* anon, const int64_t* h_e,
                       const float* h_t, const int64_t* h_d, float* h_left, float* h_right, float* prev_ts,
                       const SeqWs& w, hipStream_t st, const DropCfg& dc = DropCfg{}) {
  const int d = m->d, dm = 4 * m->d + m->d_e, H = r->hist_len, nh = r->n_head, dh = dm / nh;
  if ((int64_t)n * H >= ((int64_t)1 << 31) - 1) return TG_EUNSUPPORTED;  // slot indices are int32
  const unsigned nwg = (unsigned)std::min<int64_t>(cdiv(n, 4), 4096), nbg = (unsigned)std::min<int64_t>(n, 8192);
  hipLaunchKernelGGL(k_seq_count, dim3(nwg), dim3(256), 0, st, n, n_dev, H, h_n, w.cnt);
  hipLaunchKernelGGL(k_seq_scan, dim3(1), dim3(1024), 0, st, n, w.cnt, w.base, w.rows);
  if (w.wide)
    hipLaunchKernelGGL((k_seq_build_c<true>), dim3(nbg), dim3(256), 0, st, *m, *r, n, n_dev, nids, h_n, anon, h_e, h_t, h_d,
                       w.base, w.slot_row, w.row_slot, w.row_anon, (float4*)w.xc, w.oh, w.ohw, prev_ts);
  else
    hipLaunchKernelGGL((k_seq_build_c<false>), dim3(nbg), dim3(256), 0, st, *m, *r, n, n_dev, nids, h_n, anon, h_e, h_t, h_d,
                       w.base, w.slot_row, w.row_slot, w.row_anon, (float4*)w.xc, w.oh, w.ohw, prev_ts);
  int rc;
  GemmArgs g{};
  const int off = w.wide ? 0 : 2 * d;   // first column of the value operand that can be non-zero
  const int col0 = w.wide ? 0 : 3 * d;  // first column of the compact operand
  const float* ta = w.ta;
  if (!w.wide && r->ta_cached && dc.p <= 0.f && !w.rbar) {
    ta = r->ta_cached;  // inference with fixed parameters: the caller's table (tg_seq_restarter.ta_cached)
  } else if (!w.wide) {  // T_a = anony_emb Wqk[:, 2d:3d]^T (no bias: it rides on the compact product)
    g.m_cap = H + 1; g.n = 2 * dm; g.k = d; g.a0 = ASeg{r->anony_emb, d, d, nullptr};
    g.w = r->in_proj_w + 2 * d; g.ldw = dm; g.c = w.ta; g.ldc = 2 * dm; g.alpha = 1.f; g.nbatch = 1;
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  }
  // [q | k] of the compact rows = Xc Win[0:2dm, col0:]^T + b[0:2dm]
  g = GemmArgs{};
  g.m_cap = w.rowcap; g.m_dev = w.rows; g.n = 2 * dm; g.k = w.wx; g.a0 = ASeg{w.xc, w.wx, w.wx, nullptr};
  g.w = r->in_proj_w + col0; g.ldw = dm; g.bias = r->in_proj_b; g.c = w.qk; g.ldc = 2 * dm; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  const SeqRows sr{w.qk, w.slot_row, w.row_slot, w.row_anon, w.base, ta, anon};
  if ((rc = seq_lds_attr()) != TG_OK) return rc;
#define TG_SEQ_SCORES(HP_, V2_, MIN_)                                                                               \
  hipLaunchKernelGGL((k_seq_scores<HP_, V2_>), dim3((unsigned)(n * nh)), dim3(256), sizeof(SeqLds<HP_>), st, n, H, dm, nh, \
                     sr, h_n, w.abar, n_dev, dc, w.rbar, MIN_)
  // two launches over the same grid for histories of 33 .. 64 events: the nodes with at most 32 rows (most of them: a row
  // per real event) take the small-grid blocks - 22 KB of LDS, seven blocks per CU - the others the 64-row ones
  if (H <= 64) {
    if (dh % 2 == 0) TG_SEQ_SCORES(32, true, 0);
    else TG_SEQ_SCORES(32, false, 0);
    if (H > 32) {
      if (dh % 2 == 0) TG_SEQ_SCORES(64, true, 32);
      else TG_SEQ_SCORES(64, false, 32);
    }
  } else if (dh % 2 == 0) TG_SEQ_SCORES(128, true, 0);
  else TG_SEQ_SCORES(128, false, 0);
#undef TG_SEQ_SCORES
  const SeqCols sc{(const float4*)w.xc, w.wx / 4, col0 / 4, w.wide ? nullptr : (const float4*)r->anony_emb, w.base, w.row_slot, w.row_anon};
  hipLaunchKernelGGL(k_seq_mix, dim3(flat_grid(n * nh * (dm / 4), 256)), dim3(256), 0, st, n, H, dm / 4, d / 4, nh, w.abar,
                     sc, (float4*)w.xbar, n_dev);
  // o[:, h] = Wv_h xbar_h + bv_h   (narrow form: the first 2d columns of xbar are zeros and are skipped)
  g = GemmArgs{};
  g.m_cap = n; g.m_dev = n_dev; g.n = dh; g.k = dm - off; g.a0 = ASeg{w.xbar + off, (int64_t)nh * dm, dm - off, nullptr};
  g.a0_bs = dm;
  g.w = r->in_proj_w + (int64_t)2 * dm * dm + off; g.ldw = dm; g.w_bs = (int64_t)dh * dm;
  g.bias = r->in_proj_b + 2 * dm; g.bias_bs = dh; g.c = w.o; g.ldc = dm; g.c_bs = dh; g.alpha = 1.f; g.nbatch = nh;
  if (dc.p > 0.f && w.rbar) { g.bias_rs = w.rbar; g.ld_brs = nh; }
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // relu(mean_t out_t) = relu(Wo o + bo)
  g = GemmArgs{};
  g.m_cap = n; g.m_dev = n_dev; g.n = dm; g.k = dm; g.a0 = ASeg{w.o, dm, dm, nullptr};
  g.w = r->out_proj.w; g.ldw = dm; g.bias = r->out_proj.b; g.c = w.om; g.ldc = dm; g.relu = 1; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // h(t'-) = out_fn(...)
  g = GemmArgs{};
  g.m_cap = n; g.m_dev = n_dev; g.n = d; g.k = dm; g.a0 = ASeg{w.om, dm, dm, nullptr};
  g.w = r->out_fn.w; g.ldw = dm; g.bias = r->out_fn.b; g.c = h_left; g.ldc = d; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // h(t'+) = merger(h_left, last_event_feat) where last_event_feat is all zeros: the
  // reference takes a VIEW of full_vals and zeroes it in place before use
  // (restarters.py:102-103), so only the first d columns of fc1 contribute.
  g = GemmArgs{};
  g.m_cap = n; g.m_dev = n_dev; g.n = d; g.k = d; g.a0 = ASeg{h_left, d, d, nullptr};
  g.w = r->fc1.w; g.ldw = dm; g.bias = r->fc1.b; g.c = w.t2; g.ldc = d; g.relu = 1; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  if (dc.p > 0.f)
    hipLaunchKernelGGL(k_dropout_rows, dim3(flat_grid(n * d, 256)), dim3(256), 0, st, n, n_dev, d, w.t2, dc,
                       (uint32_t)DROP_SEQ_MERGER);
  g = GemmArgs{};
  g.m_cap = n; g.m_dev = n_dev; g.n = d; g.k = d; g.a0 = ASeg{w.t2, d, d, nullptr};
  g.w = r->fc2.w; g.ldw = d; g.bias = r->fc2.b; g.c = h_right; g.ldc = d; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // NB: the reference's invalid_rows mask can never fire (mask[:, -1] is cleared before
  // .all(1), restarters.py:86-88), so nothing is zeroed here either.
  return check_launch("tg_restart_seq_fwd");
}

// ---- the mutual-learning half of the training step -----------------------------------
struct MutualWs {
  double *ts2, *tu;
  int64_t *uniq, *index, *h_n, *h_e, *h_d, *anon;
  float *h_t, *sl, *sr, *prev_ts, *dsl, *dsr, *dt2, *dom, *dO, *dOm, *dxbar, *dabar, *dqk, *dqk_last, *dXs, *dTaT, *acc, *bpart, *cpart;
  int bb_blocks;  // blocks of k_seq_build_bwd = rows of bpart
  float* part2;   // split partials of the tabulated block's two small weight-gradient products
  size_t part2_floats;
  int32_t *count, *counts2;
  uint8_t* valid;
  void* sel_ws;
  size_t sel_bytes;
  SeqWs seq;
};

static bool carve_mutual(const tg_model* m, const tg_seq_restarter* r, int64_t B, Carver& cv, MutualWs& w) {
  const int64_t n = 2 * B;
  const size_t d = m->d;
  w.ts2 = cv.take<double>(n);
  w.tu = cv.take<double>(n);
  w.uniq = cv.take<int64_t>(n);
  w.index = cv.take<int64_t>(n);
  w.count = cv.take<int32_t>(4);
  w.counts2 = cv.take<int32_t>(4);
  w.sl = cv.take<float>(n * d);
  w.sr = cv.take<float>(n * d);
  w.dsl = cv.take<float>(n * d);
  w.dsr = cv.take<float>(n * d);
  w.valid = cv.take<uint8_t>(2 * n);
  w.acc = cv.take<float>(4);
  w.sel_bytes = tg_select_latest_workspace_bytes(n, m->n_nodes);
  w.sel_ws = cv.take<char>(w.sel_bytes);
  if (r) {
    const size_t H = r->hist_len, dm = 4 * d + m->d_e, nh = r->n_head;
    w.h_n = cv.take<int64_t>(n * H);
    w.h_e = cv.take<int64_t>(n * H);
    w.h_d = cv.take<int64_t>(n * H);
    w.anon = cv.take<int64_t>(n * H);
    w.h_t = cv.take<float>(n * H);
    w.prev_ts = cv.take<float>(n);
    w.dt2 = cv.take<float>(n * d);
    w.dom = cv.take<float>(n * dm);
    w.dO = cv.take<float>(n * dm);
    w.dOm = cv.take<float>(n * nh * dm);
    w.dxbar = cv.take<float>(n * nh * dm);
    w.dabar = cv.take<float>(n * nh * H);
    if (!carve_seq(m, r, n, cv, w.seq, true)) return false;
    w.dqk = cv.take<float>((size_t)w.seq.rowcap * 2 * dm);
    w.dqk_last = cv.take<float>(n * 2 * dm);
    w.dXs = cv.take<float>((size_t)w.seq.rowcap * (w.seq.wide ? 2 : 1) * d);
    w.dTaT = w.seq.wide ? nullptr : cv.take<float>(2 * dm * (size_t)w.seq.ohw);  // d T_a [ohw, 2 dm]
    w.cpart = cv.take<float>((size_t)16 * 2 * dm);  // partial column sums of the last slots' gradients
    w.part2_floats = w.seq.wide ? 0 : (size_t)16 * std::max((size_t)w.seq.ohw * (2 * dm + 1), 2 * dm * (d + 1));
    w.part2 = w.seq.wide ? nullptr : cv.take<float>(w.part2_floats);
    w.bb_blocks = (int)std::min<int64_t>(cdiv(n * (int64_t)H, 32), 1024);
    w.bpart = cv.take<float>((size_t)w.bb_blocks * (2 * d + (H + 1) * d));
  }
  return cv.ok;
}

size_t mutual_ws_bytes(const tg_model* m, const tg_seq_restarter* r, int64_t B) {
  return carve_bytes([&](Carver& cv) {
           MutualWs w{};
           carve_mutual(m, r, B, cv, w);
         }) +
         256;
}

// Mutual loss and its gradients (tiger.py:574-590), after STEP 4/5 produced the targets
// h_prev_left / h_prev_right.  r != NULL: SeqRestarter; else StaticRestarter tables.
int mutual_step(const tg_model* m, const tg_tcsr* g, const tg_step_io* sio, const StepWs& sw,
                const tg_seq_restarter* r, const tg_seq_restarter* gr, const float* st_left, const float* st_right,
                float* g_left, float* g_right, float* loss_out, int32_t* flag_out, float* part, size_t part_floats,
                void* ws, size_t ws_bytes, const DropCfg& dc, hipStream_t st, int phase, SideCtx* side) {
  // phase 1: the restarter's forward (restart data, histories, surrogate rows) - reads the batch, the graph and the
  // restarter's parameters only, so the training step runs it beside the contrast half on a second stream;
  // phase 2: loss and backward (needs the targets of STEP 4/5); 3: both
  if (!sio->h_prev_left || !sio->h_prev_right) return TG_EINVAL;
  if (r && (!seq_ok(m, r) || r->hist_len > 128 || !gr)) return TG_EUNSUPPORTED;
  if (!r && (!st_left || !st_right || !g_left || !g_right)) return TG_EINVAL;
  const int64_t B = sio->B, n = 2 * B;
  const int d = m->d;
  Carver cv(ws, ws_bytes);
  MutualWs w{};
  if (!carve_mutual(m, r, B, cv, w)) return TG_EWORKSPACE;
  int rc;
  const int H = r ? r->hist_len : 1;
  auto F = [](const float* p) { return const_cast<float*>(p); };
  if (phase & 1) {
  // ---- restart data (data_loader.py:133-165): latest occurrence of every positive node, float64 times
  hipLaunchKernelGGL(k_restart_queries, dim3(flat_grid(n, 256)), dim3(256), 0, st, B, sio->ts, sio->offset_dev, w.ts2);
  if ((rc = tg_select_latest(n, sw.nids3, w.ts2, 1, m->n_nodes, w.uniq, w.index, w.count, w.sel_ws, w.sel_bytes,
                             (void*)st)) != TG_OK)
    return rc;
  hipLaunchKernelGGL(k_restart_pad, dim3(flat_grid(n, 256)), dim3(256), 0, st, n, w.count, H, w.uniq, w.index, w.ts2,
                     w.tu, w.counts2, w.acc);
  if (!r) {
    hipLaunchKernelGGL(k_static_rows, dim3(flat_grid(n * d, 256)), dim3(256), 0, st, n, w.counts2, d, w.uniq, st_left,
                       st_right, w.sl, w.sr);
  } else {
    if ((rc = tg_sample_recent_edges(g, n, w.uniq, w.tu, H, w.h_n, w.h_e, w.h_t, w.h_d, nullptr, (void*)st)) != TG_OK)
      return rc;
    if ((rc = tg_anonymized_reindex(n, H, w.h_n, w.anon, (void*)st)) != TG_OK) return rc;
    if ((rc = seq_forward(m, r, n, w.counts2, w.uniq, w.h_n, w.anon, w.h_e, w.h_t, w.h_d, w.sl, w.sr, w.prev_ts, w.seq,
                          st, dc)) != TG_OK)
      return rc;
  }
  }
  if (!(phase & 2)) return check_launch("mutual_step(forward)");
  hipLaunchKernelGGL(k_mutual_a, dim3(std::min<unsigned>(flat_grid(2 * n, 4), 128)), dim3(256), 0, st, n, w.counts2, d,
                     w.index, sio->h_prev_left, sio->h_prev_right, w.sl, w.sr, w.valid, w.acc);
  hipLaunchKernelGGL(k_mutual_b, dim3(flat_grid(2 * n * d, 256)), dim3(256), 0, st, n, w.counts2, d, w.index,
                     sio->h_prev_left, sio->h_prev_right, w.sl, w.sr, w.valid, w.acc, w.dsl, w.dsr, loss_out, flag_out);
  if (!r) {
    hipLaunchKernelGGL(k_static_rows_bwd, dim3(flat_grid(n * d, 256)), dim3(256), 0, st, n, w.counts2, d, w.uniq, w.dsl,
                       w.dsr, g_left, g_right);
    return check_launch("mutual_step(static)");
  }
  // ---- SeqRestarter backward
  const int dm = 4 * d + m->d_e, nh = r->n_head;
  const int32_t* n_dev = w.counts2;
  const SeqWs& q = w.seq;
  const int32_t* rows_dev = q.rows;
  const int off = q.wide ? 0 : 2 * d, col0 = q.wide ? 0 : 3 * d;
  TnArgs tn{};
  GemmArgs ga{};
  // The weight-gradient products depend on the gradient chain but nothing depends on them: they are collected and launched
  // as two groups (gemm_tn_group_launch: one launch + one reduction each) - the value-side ones behind the per-head products,
  // the q / k projection's behind the score backward - on the side lane when there is one (a fork each, one join before the
  // parameter sums of the build pass), while the chain goes on on the main stream.
  const size_t lfull = (size_t)((H + 1) * d) * sizeof(float);
  const int use_lds = lfull <= 60 * 1024;  // (else the build pass adds to d anony_emb with atomics while it runs: one stream)
  const bool side_ok = side != nullptr && use_lds && side->used + 3 <= side->n;
  std::vector<TnArgs> wq, late_heads;
  auto wgrad = [&](const TnArgs& a) {
    wq.push_back(a);
    return (int)TG_OK;
  };
  hipStream_t ws_st = st;  // stream of the last flushed group
  auto flush = [&]() {
    ws_st = st;
    if (side_ok) {
      if (!side->after_main(st)) {
        set_hip_error(hipGetLastError(), "mutual_step lane fork");
        return (int)TG_EHIP;
      }
      ws_st = side->s;
    }
    int rcf = TG_OK;
    for (size_t o = 0; o < wq.size() && rcf == TG_OK; o += TN_GROUP_MAX)
      rcf = gemm_tn_group_launch(wq.data() + o, (int)std::min<size_t>(TN_GROUP_MAX, wq.size() - o), part, part_floats, ws_st);
    wq.clear();
    return rcf;
  };
  auto tn_base = [&](int64_t cap, const int32_t* md) {
    TnArgs t{};
    t.m_cap = cap; t.m_dev = md; t.alpha = 1.f; t.accumulate = 1; t.nbatch = 1; t.part = part; t.part_floats = part_floats;
    t.bias_accumulate = 1;
    return t;
  };
  // merger fc2 / fc1 (only the h_left columns of fc1 ever see a non-zero input)
  tn = tn_base(n, n_dev);
  tn.n = d; tn.k = d; tn.y = w.dsr; tn.ldy = d; tn.x0 = ASeg{q.t2, d, d, nullptr};
  tn.out = F(gr->fc2.w); tn.ldo = d; tn.bias_out = F(gr->fc2.b);
  if ((rc = wgrad(tn)) != TG_OK) return rc;
  ga = GemmArgs{};
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = d; ga.k = d; ga.a0 = ASeg{w.dsr, d, d, nullptr};
  ga.w = r->fc2.w; ga.ldw = d; ga.w_kmajor = 1; ga.c = w.dt2; ga.ldc = d; ga.nbatch = 1;
  ga.alpha = dc.p > 0.f ? dc.scale : 1.f;  // q.t2 is the dropped activation: > 0 iff kept and positive
  ga.relu_mask = q.t2; ga.ld_mask = d;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  tn = tn_base(n, n_dev);
  tn.n = d; tn.k = d; tn.y = w.dt2; tn.ldy = d; tn.x0 = ASeg{w.sl, d, d, nullptr};
  tn.out = F(gr->fc1.w); tn.ldo = dm; tn.bias_out = F(gr->fc1.b);
  if ((rc = wgrad(tn)) != TG_OK) return rc;
  ga = GemmArgs{};  // d h_left += dt2 fc1[:, :d]
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = d; ga.k = d; ga.a0 = ASeg{w.dt2, d, d, nullptr};
  ga.w = r->fc1.w; ga.ldw = dm; ga.w_kmajor = 1; ga.c = w.dsl; ga.ldc = d; ga.alpha = 1.f; ga.nbatch = 1; ga.accumulate = 1;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  // out_fn
  tn = tn_base(n, n_dev);
  tn.n = d; tn.k = dm; tn.y = w.dsl; tn.ldy = d; tn.x0 = ASeg{q.om, dm, dm, nullptr};
  tn.out = F(gr->out_fn.w); tn.ldo = dm; tn.bias_out = F(gr->out_fn.b);
  if ((rc = wgrad(tn)) != TG_OK) return rc;
  ga = GemmArgs{};
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = dm; ga.k = d; ga.a0 = ASeg{w.dsl, d, d, nullptr};
  ga.w = r->out_fn.w; ga.ldw = dm; ga.w_kmajor = 1; ga.c = w.dom; ga.ldc = dm; ga.alpha = 1.f; ga.nbatch = 1;
  ga.relu_mask = q.om; ga.ld_mask = dm;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  // out_proj
  tn = tn_base(n, n_dev);
  tn.n = dm; tn.k = dm; tn.y = w.dom; tn.ldy = dm; tn.x0 = ASeg{q.o, dm, dm, nullptr};
  tn.out = F(gr->out_proj.w); tn.ldo = dm; tn.bias_out = F(gr->out_proj.b);
  if ((rc = wgrad(tn)) != TG_OK) return rc;
  ga = GemmArgs{};
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = dm; ga.k = dm; ga.a0 = ASeg{w.dom, dm, dm, nullptr};
  ga.w = r->out_proj.w; ga.ldw = dm; ga.w_kmajor = 1; ga.c = w.dO; ga.ldc = dm; ga.alpha = 1.f; ga.nbatch = 1;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  // value projection per head.  dh = dm / nh need not be a multiple of 4 (d = 172: dh = 430), so the
  // head slices of dO cannot be addressed as aligned sub-matrices; instead each head uses a copy of
  // dO with the other heads' columns zeroed and full-width (K = dm) products.  Narrow form: the first 2d input columns
  // are zeros - no weight gradient there, and their input gradient is not needed.
  hipLaunchKernelGGL(k_head_mask, dim3(flat_grid(n * dm, 256)), dim3(256), 0, st, n, n_dev, dm, nh, w.dO, w.dOm);
  for (int h = 0; h < nh; ++h) {
    const float* dOh = w.dOm + (int64_t)h * n * dm;
    tn = tn_base(n, n_dev);
    tn.n = dm; tn.k = dm - off; tn.y = dOh; tn.ldy = dm;
    tn.x0 = ASeg{q.xbar + (int64_t)h * dm + off, (int64_t)nh * dm, dm - off, nullptr};
    tn.out = F(gr->in_proj_w) + (int64_t)2 * dm * dm + off; tn.ldo = dm; tn.bias_out = F(gr->in_proj_b) + 2 * dm;
    if (dc.p > 0.f) { tn.bias_rs = q.rbar; tn.ld_brs = nh; tn.brs_col = h; }
    // (all heads add to the same rows - each to its own, zeros to the others': one of them per group, the rest behind it)
    if (h == 0) {
      if ((rc = wgrad(tn)) != TG_OK) return rc;
    } else {
      late_heads.push_back(tn);
    }
  }
  // d xbar of all heads as ONE batched product (same weights, the heads' masked copies of dO, their column blocks of dxbar):
  // a head alone is 150 blocks of work on 256 CUs
  ga = GemmArgs{};
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = dm - off; ga.k = dm; ga.a0 = ASeg{w.dOm, dm, dm, nullptr}; ga.a0_bs = (int64_t)n * dm;
  ga.w = r->in_proj_w + (int64_t)2 * dm * dm + off; ga.ldw = dm; ga.w_kmajor = 1; ga.w_bs = 0;
  ga.c = w.dxbar + off; ga.ldc = (int64_t)nh * dm; ga.c_bs = dm; ga.alpha = 1.f; ga.nbatch = nh;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  if ((rc = flush()) != TG_OK) return rc;
  for (const TnArgs& a : late_heads)
    if ((rc = gemm_tn_launch(a, ws_st)) != TG_OK) return rc;
  // value mix and attention scores
  const SeqCols sc{(const float4*)q.xc, q.wx / 4, col0 / 4, q.wide ? nullptr : (const float4*)r->anony_emb, q.base, q.row_slot, q.row_anon};
  hipLaunchKernelGGL(k_seq_mix_bwd, dim3(flat_grid(n * nh * H, 4)), dim3(256), 0, st, n, n_dev, rows_dev, H, dm / 4, d / 4, nh,
                     (const float4*)w.dxbar, sc, w.dabar, dc.p > 0.f ? w.dO : (const float*)nullptr,
                     r->in_proj_b + 2 * dm);
  const SeqRows sr{q.qk, q.slot_row, q.row_slot, q.row_anon, q.base, q.ta, w.anon};
#define TG_SEQ_SCORES_BWD(HP_, V2_, MIN_, ST_)                                                                        \
  hipLaunchKernelGGL((k_seq_scores_bwd<HP_, V2_>), dim3((unsigned)(n * nh)), dim3(256), sizeof(SeqLds<HP_>), ST_, n, n_dev, H, \
                     dm, nh, sr, w.h_n, w.dabar, w.dqk, w.dqk_last, dc, MIN_)
  const bool v2 = (dm / nh) % 2 == 0;
  if (H <= 64) {
    // histories of 33 .. 64 events: the two row classes are two launches over the same grid that write disjoint rows - the
    // many short blocks of the 32-row class on the lane (when there is one) beside the few long ones of the 64-row class
    // (TG_SEQ_BWD_SPLIT=0: one after the other on the main stream)
    static const int split_knob = getenv("TG_SEQ_BWD_SPLIT") ? atoi(getenv("TG_SEQ_BWD_SPLIT")) : 1;  // tuning knob
    hipStream_t s32 = st;
    bool forked = false;
    if (H > 32 && side_ok && split_knob && side->used + 5 <= side->n && side->after_main(st)) {
      s32 = side->s;
      forked = true;
    }
    if (v2) TG_SEQ_SCORES_BWD(32, true, 0, s32);
    else TG_SEQ_SCORES_BWD(32, false, 0, s32);
    if (H > 32) {
      if (v2) TG_SEQ_SCORES_BWD(64, true, 32, st);
      else TG_SEQ_SCORES_BWD(64, false, 32, st);
    }
    if (forked) {  // join: the column sums and the products below read both classes' rows
      if (hipEventRecord(side->ev[side->used], side->s) != hipSuccess || hipStreamWaitEvent(st, side->ev[side->used], 0) != hipSuccess) {
        set_hip_error(hipGetLastError(), "mutual_step (score backward join)");
        return TG_EHIP;
      }
      ++side->used;
    }
  } else if (v2) TG_SEQ_SCORES_BWD(128, true, 0, st);
  else TG_SEQ_SCORES_BWD(128, false, 0, st);
#undef TG_SEQ_SCORES_BWD
  // row 0 (the last event of every node) collects the last slots' gradients
  if ((rc = colsum_launch(n, n_dev, 2 * dm, w.dqk_last, 2 * dm, 1.f, w.dqk, 0, w.cpart, (size_t)16 * 2 * dm, st)) != TG_OK) return rc;
  // q/k projection: weight columns [col0, dm), bias
  tn = tn_base(q.rowcap, rows_dev);
  tn.n = 2 * dm; tn.k = q.wx; tn.y = w.dqk; tn.ldy = 2 * dm; tn.x0 = ASeg{q.xc, q.wx, q.wx, nullptr};
  tn.out = F(gr->in_proj_w) + col0; tn.ldo = dm; tn.bias_out = F(gr->in_proj_b);
  if ((rc = wgrad(tn)) != TG_OK) return rc;
  if ((rc = flush()) != TG_OK) return rc;  // the q / k projection's weight gradient: on the lane when there is one
  if (!q.wide) {
    // the tabulated anony_emb block, T_a = anony_emb Wa^T with Wa = in_proj_w[0:2dm, 2d:3d]:  d T_a = onehot^T dqk
    // ([ohw, 2 dm]), then  d anony_emb += d T_a Wa  and  d Wa += d T_a^T anony_emb  (rows past H of d T_a are zeros: no slot
    // carries such an id; the two products read its first H + 1 rows only).  On the main stream, with partials of their
    // own: the lane is the longer branch.
    tn = tn_base(q.rowcap, rows_dev);
    tn.part = w.part2; tn.part_floats = w.part2_floats;
    tn.accumulate = 0; tn.bias_accumulate = 0;
    tn.n = q.ohw; tn.k = 2 * dm; tn.y = q.oh; tn.ldy = q.ohw; tn.x0 = ASeg{w.dqk, 2 * dm, 2 * dm, nullptr};
    tn.out = w.dTaT; tn.ldo = 2 * dm;
    if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
    ga = GemmArgs{};
    ga.m_cap = H + 1; ga.n = d; ga.k = 2 * dm; ga.a0 = ASeg{w.dTaT, 2 * dm, 2 * dm, nullptr};
    ga.w = r->in_proj_w + 2 * d; ga.ldw = dm; ga.w_kmajor = 1; ga.c = F(gr->anony_emb); ga.ldc = d; ga.alpha = 1.f;
    ga.nbatch = 1; ga.accumulate = 1;
    if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
    tn = tn_base(H + 1, nullptr);
    tn.part = w.part2; tn.part_floats = w.part2_floats;
    tn.n = 2 * dm; tn.k = d; tn.y = w.dTaT; tn.ldy = 2 * dm; tn.x0 = ASeg{r->anony_emb, d, d, nullptr};
    tn.out = F(gr->in_proj_w) + 2 * d; tn.ldo = dm;
    if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
  }
  // input-row gradients, only for the column blocks that carry parameters (main stream, beside the products above)
  const int ldx = q.wide ? 2 * d : d;
  ga = GemmArgs{};
  ga.m_cap = q.rowcap; ga.m_dev = rows_dev; ga.n = d; ga.k = 2 * dm; ga.a0 = ASeg{w.dqk, 2 * dm, 2 * dm, nullptr};
  ga.w = r->in_proj_w + 3 * d + m->d_e; ga.ldw = dm; ga.w_kmajor = 1; ga.c = w.dXs + (q.wide ? d : 0); ga.ldc = ldx;
  ga.alpha = 1.f; ga.nbatch = 1;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  if (q.wide) {
    ga.w = r->in_proj_w + 2 * d; ga.c = w.dXs;
    if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  }
  const int nl = use_lds ? (H + 1) * d : 0;
  hipLaunchKernelGGL(k_seq_build_bwd, dim3((unsigned)w.bb_blocks), dim3(256), use_lds ? lfull : (size_t)16, st, *m, *r, n,
                     n_dev, w.anon, w.h_t, q.slot_row, q.row_slot, w.dXs, q.wide ? 1 : 0, q.abar, w.dxbar, use_lds,
                     F(gr->anony_emb), w.bpart);
  if (side) {  // the lane's products are done before the parameter sums (both add to d anony_emb) and before the caller goes on
    if (side->used >= side->n || hipEventRecord(side->ev[side->used], side->s) != hipSuccess ||
        hipStreamWaitEvent(st, side->ev[side->used], 0) != hipSuccess) {
      set_hip_error(hipGetLastError(), "mutual_step lane join");
      return TG_EHIP;
    }
    ++side->used;
  }
  hipLaunchKernelGGL(k_seq_build_reduce, dim3((unsigned)cdiv(2 * d + nl, 64)), dim3(1024), 0, st, w.bb_blocks, d, nl, w.bpart,
                     F(gr->te_freq), F(gr->te_phase), F(gr->anony_emb));
  return check_launch("mutual_step(seq)");
}

}  // namespace tg

using namespace tg;

extern "C" size_t tg_restart_seq_workspace_bytes(const tg_model* m, const tg_seq_restarter* r, int64_t n) {
  if (!seq_ok(m, r) || n < 0) return 0;
  return carve_bytes([&](Carver& cv) {
           SeqWs w{};
           carve_seq(m, r, n, cv, w, true);
         }) +
         64;
}

extern "C" int tg_restart_seq_fwd(const tg_model* m, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                                  const int64_t* h_n, const int64_t* anon, const int64_t* h_e, const float* h_t,
                                  const int64_t* h_d, float* h_left, float* h_right, float* prev_ts, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!seq_ok(m, r) || n < 0) return TG_EINVAL;
  if (r->hist_len > 128) return TG_EUNSUPPORTED;
  if (n == 0) return TG_OK;
  if (!nids || !h_n || !anon || !h_e || !h_t || !h_d || !h_left || !h_right || !prev_ts) return TG_EINVAL;
  Carver cv(ws, ws_bytes);
  SeqWs w{};
  if (!ws || !carve_seq(m, r, n, cv, w, false)) return TG_EWORKSPACE;
  return seq_forward(m, r, n, nullptr, nids, h_n, anon, h_e, h_t, h_d, h_left, h_right, prev_ts, w, as_stream(stream));
}

namespace tg {
// query times of a list restart: every node at the batch's earliest time, float32-rounded (restarters.py:69-70)
__global__ void k_restart_times(int64_t n, const float* __restrict__ t_dev, double* __restrict__ tu) {
  const double t = (double)*t_dev;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) tu[i] = t;
}
// several lists (each with its own earliest time) as one: ids concatenated in list order, the query time of every entry
struct SegLists {
  const int64_t* ids[TG_RESTART_MAX_LISTS];
  const float* t[TG_RESTART_MAX_LISTS];
  int64_t off[TG_RESTART_MAX_LISTS + 1];
  int n;
};
__global__ void k_concat_lists(SegLists sl, int64_t* __restrict__ ids, double* __restrict__ tu) {
  const int64_t total = sl.off[sl.n];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int s = 0;
    while (s + 1 < sl.n && i >= sl.off[s + 1]) ++s;
    ids[i] = sl.ids[s][i - sl.off[s]];
    tu[i] = (double)*sl.t[s];
  }
}
// the StaticRestarter over several lists (restarters.py:262-277): both table rows of every listed node and the time of its
// last event strictly before the list's (float32) time - 0 when there is none
__global__ void k_static_lists(tg_tcsr g, SegLists sl, int d4, const float4* __restrict__ left, const float4* __restrict__ right,
                               int64_t* __restrict__ ids, float4* __restrict__ hl, float4* __restrict__ hr,
                               float* __restrict__ pt) {
  const int64_t total = sl.off[sl.n] * d4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / d4;
    const int c = (int)(t - i * d4);
    int s = 0;
    while (s + 1 < sl.n && i >= sl.off[s + 1]) ++s;
    const int64_t v = sl.ids[s][i - sl.off[s]];
    hl[t] = left[v * d4 + c];
    hr[t] = right[v * d4 + c];
    if (c == 0) {
      int64_t start;
      const int64_t end = prefix_end(g, v, (double)*sl.t[s], &start);
      ids[i] = v;
      pt[i] = end > start ? (float)g.ts[end - 1] : 0.f;
    }
  }
}
struct ListWs {
  double* tu;
  int64_t *h_n, *h_e, *h_d, *anon;
  float *h_t, *hl, *hr, *pt;
  SeqWs seq;
};
static bool carve_list(const tg_model* m, const tg_seq_restarter* r, int64_t n, Carver& cv, ListWs& w, bool train = false) {
  const size_t H = r->hist_len, d = m->d;
  w.tu = cv.take<double>(n);
  w.h_n = cv.take<int64_t>(n * H);
  w.h_e = cv.take<int64_t>(n * H);
  w.h_d = cv.take<int64_t>(n * H);
  w.anon = cv.take<int64_t>(n * H);
  w.h_t = cv.take<float>(n * H);
  w.hl = cv.take<float>(n * d);
  w.hr = cv.take<float>(n * d);
  w.pt = cv.take<float>(n);
  return carve_seq(m, r, n, cv, w.seq, train);  // (train() mode: the dropout form keeps two more buffers)
}
}  // namespace tg

extern "C" size_t tg_restart_seq_list_workspace_bytes(const tg_model* m, const tg_seq_restarter* r, int64_t n) {
  if (!seq_ok(m, r) || n < 0) return 0;
  return carve_bytes([&](Carver& cv) {
           ListWs w{};
           carve_list(m, r, n, cv, w, true);  // (the larger of the two forms)
         }) +
         64;
}

extern "C" int tg_restart_seq_list(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t n,
                                   const int64_t* nids, const float* t_dev, void* ws, size_t ws_bytes, void* stream) {
  return tg_restart_seq_list_dev(m, g, r, n, nids, nullptr, t_dev, ws, ws_bytes, stream);
}

// histories at the batch's earliest time, anonymised ids and the restarter's forward of a device-resident list: the rows go to
// (hl, hr, pt) - the caller's, or the workspace's own when NULL; `w` tells where they are
static int list_forward(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                        const int32_t* n_dev, const float* t_dev, float* hl, float* hr, float* pt, void* ws, size_t ws_bytes,
                        void* stream, ListWs& w, const SegLists* segs = nullptr, int64_t* ids_out = nullptr,
                        const DropCfg& dc = DropCfg{}) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!seq_ok(m, r) || !g || n < 0) return TG_EINVAL;
  if (r->hist_len > 128) return TG_EUNSUPPORTED;
  if (n == 0) return TG_OK;
  if (segs ? !ids_out : (!nids || !t_dev)) return TG_EINVAL;
  Carver cv(ws, ws_bytes);
  if (!ws || !carve_list(m, r, n, cv, w, dc.p > 0.f)) return TG_EWORKSPACE;
  if (hl) { w.hl = hl; w.hr = hr; w.pt = pt; }
  hipStream_t st = as_stream(stream);
  const int H = r->hist_len;
  int rc;
  if (segs) {
    hipLaunchKernelGGL(k_concat_lists, dim3(flat_grid(n, 256)), dim3(256), 0, st, *segs, ids_out, w.tu);
    nids = ids_out;
  } else {
    hipLaunchKernelGGL(k_restart_times, dim3(flat_grid(n, 256)), dim3(256), 0, st, n, t_dev, w.tu);
  }
  if ((rc = tg_sample_recent_edges(g, n, nids, w.tu, H, w.h_n, w.h_e, w.h_t, w.h_d, nullptr, stream)) != TG_OK) return rc;
  if ((rc = tg_anonymized_reindex(n, H, w.h_n, w.anon, stream)) != TG_OK) return rc;
  // n_dev: the launches are sized for n (a capacity), the first *n_dev entries of the list are live - the entries behind them
  // must be valid node ids (their histories are sampled and thrown away); nothing is written for them
  return seq_forward(m, r, n, n_dev, nids, w.h_n, w.anon, w.h_e, w.h_t, w.h_d, w.hl, w.hr, w.pt, w.seq, st, dc);
}

namespace tg {
__global__ void k_rng_tick_list(uint64_t* rng) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;
}
}  // namespace tg

extern "C" int tg_restart_seq_list_train(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t n,
                                         const int64_t* nids, const float* t_dev, float dropout_p, uint64_t* rng, void* ws,
                                         size_t ws_bytes, void* stream) {
  if (dropout_p < 0.f || dropout_p >= 1.f || (dropout_p > 0.f && !rng)) return TG_EINVAL;
  ListWs w{};
  const DropCfg dc = make_drop(dropout_p, rng);
  const int rc = list_forward(m, g, r, n, nids, nullptr, t_dev, nullptr, nullptr, nullptr, ws, ws_bytes, stream, w, nullptr,
                              nullptr, dc);
  if (rc != TG_OK || n == 0) return rc;
  if (dc.p > 0.f) hipLaunchKernelGGL(k_rng_tick_list, dim3(1), dim3(64), 0, as_stream(stream), rng);
  return restart_apply_dev(m, n, nids, w.hl, w.hr, w.pt, nullptr, as_stream(stream));
}

extern "C" int tg_restart_seq_list_dev(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t n,
                                       const int64_t* nids, const int32_t* n_dev, const float* t_dev, void* ws,
                                       size_t ws_bytes, void* stream) {
  ListWs w{};
  const int rc = list_forward(m, g, r, n, nids, n_dev, t_dev, nullptr, nullptr, nullptr, ws, ws_bytes, stream, w);
  if (rc != TG_OK || n == 0) return rc;
  return restart_apply_dev(m, n, nids, w.hl, w.hr, w.pt, n_dev, as_stream(stream));
}

extern "C" int tg_restart_seq_list_fwd(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t n,
                                       const int64_t* nids, const int32_t* n_dev, const float* t_dev, float* h_left,
                                       float* h_right, float* prev_ts, void* ws, size_t ws_bytes, void* stream) {
  if (n > 0 && (!h_left || !h_right || !prev_ts)) return TG_EINVAL;
  ListWs w{};
  return list_forward(m, g, r, n, nids, n_dev, t_dev, h_left, h_right, prev_ts, ws, ws_bytes, stream, w);
}

extern "C" int tg_restart_seq_lists_fwd(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int32_t n_lists,
                                        const int64_t* const* lists, const int64_t* counts, const float* const* t_dev,
                                        int64_t* ids_out, float* h_left, float* h_right, float* prev_ts, void* ws,
                                        size_t ws_bytes, void* stream) {
  if (n_lists <= 0 || n_lists > TG_RESTART_MAX_LISTS || !lists || !counts || !t_dev) return TG_EINVAL;
  SegLists sl{};
  for (int j = 0; j < n_lists; ++j) {
    if (counts[j] < 0 || (counts[j] > 0 && (!lists[j] || !t_dev[j]))) return TG_EINVAL;
    if (counts[j] == 0) continue;  // (an empty list takes no segment)
    sl.ids[sl.n] = lists[j];
    sl.t[sl.n] = t_dev[j];
    sl.off[sl.n + 1] = sl.off[sl.n] + counts[j];
    ++sl.n;
  }
  const int64_t n = sl.off[sl.n];
  if (n > 0 && (!ids_out || !h_left || !h_right || !prev_ts)) return TG_EINVAL;
  ListWs w{};
  return list_forward(m, g, r, n, nullptr, nullptr, nullptr, h_left, h_right, prev_ts, ws, ws_bytes, stream, w, &sl, ids_out);
}

extern "C" int tg_restart_static_lists_fwd(const tg_model* m, const tg_tcsr* g, const float* static_left,
                                           const float* static_right, int32_t n_lists, const int64_t* const* lists,
                                           const int64_t* counts, const float* const* t_dev, int64_t* ids_out, float* h_left,
                                           float* h_right, float* prev_ts, void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;
  if (!m || !g || m->d <= 0 || (m->d % 4) || !static_left || !static_right) return TG_EINVAL;
  if (n_lists <= 0 || n_lists > TG_RESTART_MAX_LISTS || !lists || !counts || !t_dev) return TG_EINVAL;
  SegLists sl{};
  for (int j = 0; j < n_lists; ++j) {
    if (counts[j] < 0 || (counts[j] > 0 && (!lists[j] || !t_dev[j]))) return TG_EINVAL;
    if (counts[j] == 0) continue;
    sl.ids[sl.n] = lists[j];
    sl.t[sl.n] = t_dev[j];
    sl.off[sl.n + 1] = sl.off[sl.n] + counts[j];
    ++sl.n;
  }
  const int64_t n = sl.off[sl.n];
  if (n == 0) return TG_OK;
  if (!ids_out || !h_left || !h_right || !prev_ts) return TG_EINVAL;
  hipLaunchKernelGGL(k_static_lists, dim3(flat_grid(n * (m->d / 4), 256)), dim3(256), 0, as_stream(stream), *g, sl, m->d / 4,
                     (const float4*)static_left, (const float4*)static_right, ids_out, (float4*)h_left, (float4*)h_right, prev_ts);
  return check_launch("tg_restart_static_lists_fwd");
}

namespace tg {
__global__ void k_rng_tick_r(uint64_t* rng) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;
}
}  // namespace tg

extern "C" int tg_restart_seq_fwd_train(const tg_model* m, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                                        const int64_t* h_n, const int64_t* anon, const int64_t* h_e, const float* h_t,
                                        const int64_t* h_d, float* h_left, float* h_right, float* prev_ts,
                                        float dropout_p, uint64_t* rng, void* ws, size_t ws_bytes, void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!seq_ok(m, r) || n < 0 || dropout_p < 0.f || dropout_p >= 1.f || (dropout_p > 0.f && !rng)) return TG_EINVAL;
  if (r->hist_len > 128) return TG_EUNSUPPORTED;
  if (n == 0) return TG_OK;
  if (!nids || !h_n || !anon || !h_e || !h_t || !h_d || !h_left || !h_right || !prev_ts) return TG_EINVAL;
  Carver cv(ws, ws_bytes);
  SeqWs w{};
  if (!ws || !carve_seq(m, r, n, cv, w, true)) return TG_EWORKSPACE;
  const DropCfg dc = make_drop(dropout_p, rng);
  hipStream_t st = as_stream(stream);
  const int rc = seq_forward(m, r, n, nullptr, nids, h_n, anon, h_e, h_t, h_d, h_left, h_right, prev_ts, w, st, dc);
  if (rc == TG_OK && dc.p > 0.f) hipLaunchKernelGGL(k_rng_tick_r, dim3(1), dim3(64), 0, st, rng);
  return rc;
}
