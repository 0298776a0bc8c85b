// Training tail on device: STEP 7 of TIGE.contrast_learning
// (tiger/model/tiger.py:257-288: hit features, score MergeLayer, BCE-with-logits), the
// backward pass of that loss through the temporal attention (temporal_agg_modules.py:29-83,
// 186-235), the GRU updater (update_modules.py:30-37) and the time encoder
// (time_encoding.py:24-26), and torch.optim.Adam (train_self_supervised.py:114,170).
//
// The backward pass follows the restructured forward of tg_model.hip (query folded through
// Wk, softmax-weighted raw key rows projected once through Wv): every dX product re-uses the
// forward GEMM kernel on the weight read k-major (no transposed copies), every dW product is
// k_gemm_tn (both operands read along the batch dimension, deterministic split reduction),
// and one gather kernel recomputes the K scores per centre, back-propagates the softmax and
// scatters the key-row gradients to the involved-node rows.
#include <algorithm>
#include <vector>

#include "tg_step.h"

namespace tg {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, TG_WAVE));
  return v;
}

__device__ __forceinline__ float dot4f(float4 a, float4 b, float acc) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}
__device__ __forceinline__ void axpy4f(float4& s, float b, float4 x) {
  s.x = fmaf(b, x.x, s.x);
  s.y = fmaf(b, x.y, s.y);
  s.z = fmaf(b, x.z, s.z);
  s.w = fmaf(b, x.w, s.w);
}

// ---------------------------------------------------------------------------------
// STEP 7 forward pieces
// ---------------------------------------------------------------------------------
// Pair rows for the score MergeLayer (tiger.py:259-279).  Row r < B is the positive pair of
// event r, row B + r the negative pair: P[r] = [x_pair | y_pair], each W = d (+K) wide.
// Hit windows (data_loader.py:60-75) are the recent-edges neighbour lists the step already
// sampled: nbrs(src) = l1[i], nbrs(dst) = l1[B+i], nbrs(neg) = l1[2B+i].
__global__ void __launch_bounds__(256) k_build_pairs(int64_t B, int d, int K, int hit_type,
                                                     const float* __restrict__ h, const int64_t* __restrict__ nids3,
                                                     const int64_t* __restrict__ l1_nids,
                                                     const float* __restrict__ hit_emb, float* __restrict__ P,
                                                     int32_t* __restrict__ hit_idx, float* __restrict__ zero2) {
  const int lane = lane_id();
  if (zero2 && blockIdx.x == 0 && threadIdx.x < 2) zero2[threadIdx.x] = 0.f;  // the loss accumulators (k_score_loss adds)
  const int W = d + (hit_type == TG_HIT_VEC ? K : 0);
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < 2 * B; r += (int64_t)gridDim.x * 4) {
    const bool neg = r >= B;
    const int64_t i = neg ? r - B : r;
    const int64_t src = nids3[i];
    const int64_t yrow = neg ? 2 * B + i : B + i;  // the other endpoint: dst or the negative dst
    const int64_t other = nids3[yrow];
    // x side: is src inside the window of the other endpoint; y side: is the other endpoint in src's window
    const bool hx = lane < K && l1_nids[yrow * K + lane] == src;
    const bool hy = lane < K && l1_nids[i * K + lane] == other;
    const unsigned long long bx = __ballot(hx), by = __ballot(hy);
    int ix = 0, iy = 0;
    if (hit_type == TG_HIT_BIN) {
      ix = bx != 0ull;
      iy = by != 0ull;
    } else if (hit_type == TG_HIT_COUNT) {
      ix = __popcll(bx);
      iy = __popcll(by);
    }
    if (lane == 0) {
      hit_idx[2 * r] = ix;
      hit_idx[2 * r + 1] = iy;
    }
    float* row = P + r * 2 * (int64_t)W;
    const bool emb = hit_type == TG_HIT_BIN || hit_type == TG_HIT_COUNT;
    for (int c = lane; c < d; c += TG_WAVE) {
      float vx = h[i * d + c], vy = h[yrow * d + c];
      if (emb) {
        vx += hit_emb[(int64_t)ix * d + c];
        vy += hit_emb[(int64_t)iy * d + c];
      }
      row[c] = vx;
      row[W + c] = vy;
    }
    if (hit_type == TG_HIT_VEC && lane < K) {
      row[d + lane] = hx ? 1.f : 0.f;
      row[W + d + lane] = hy ? 1.f : 0.f;
    }
  }
}

// logits, BCE-with-logits (mean over 2B), d(loss)/d(logit), fc2 gradients, and dT1 in place:
// T1[r] <- (T1[r] > 0) * dscore_r * w2.   NVS * 64 >= d.
constexpr int NVS = 8;
__global__ void __launch_bounds__(256) k_score_loss(int64_t B, int d, float* __restrict__ T1,
                                                    const float* __restrict__ w2, const float* __restrict__ b2,
                                                    float* __restrict__ pos_scores, float* __restrict__ neg_scores,
                                                    float* __restrict__ loss_out, float* __restrict__ dw2,
                                                    float* __restrict__ db2, DropCfg dc) {
  __shared__ float red[4][NVS * 64 + 2];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const uint64_t dkey = drop_key(dc);
  float acc[NVS];
#pragma unroll
  for (int v = 0; v < NVS; ++v) acc[v] = 0.f;
  float loss = 0.f, dbs = 0.f;
  const float inv = 1.f / (float)(2 * B);
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < 2 * B; r += (int64_t)gridDim.x * 4) {
    float* t = T1 + r * d;
    float tv[NVS], wv[NVS];
    float p = 0.f;
#pragma unroll
    for (int v = 0; v < NVS; ++v) {
      const int c = lane + v * 64;
      tv[v] = c < d ? t[c] : 0.f;
      wv[v] = c < d ? w2[c] : 0.f;
      if (dc.p > 0.f && c < d) {  // MergeLayer dropout on relu(fc1) (basic_modules.py:18); folded into w2 for the backward
        const float ms = drop_keep(dkey, DROP_SCORE, (uint64_t)r * d + c, dc.thresh) ? dc.scale : 0.f;
        tv[v] *= ms;
        wv[v] *= ms;
      }
      p = fmaf(tv[v], c < d ? w2[c] : 0.f, p);
    }
    const float s = wave_sum(p) + b2[0];
    const float y = r < B ? 1.f : 0.f;
    if (lane == 0) {
      if (r < B) {
        if (pos_scores) pos_scores[r] = s;
      } else if (neg_scores) {
        neg_scores[r - B] = s;
      }
    }
    loss += fmaxf(s, 0.f) - s * y + log1pf(expf(-fabsf(s)));
    const float ds = (1.f / (1.f + expf(-s)) - y) * inv;
    dbs += ds;
    if (dw2) {
#pragma unroll
      for (int v = 0; v < NVS; ++v) {
        const int c = lane + v * 64;
        acc[v] = fmaf(ds, tv[v], acc[v]);
        if (c < d) t[c] = tv[v] > 0.f ? ds * wv[v] : 0.f;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NVS; ++v) red[wave][v * 64 + lane] = acc[v];
  if (lane == 0) {
    red[wave][NVS * 64] = loss;
    red[wave][NVS * 64 + 1] = dbs;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < NVS * 64 + 2; c += 256) {
    const float s = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (c < d) {
      if (dw2) atomicAdd(dw2 + c, s);
    } else if (c == NVS * 64) {
      atomicAdd(loss_out, s * inv);
    } else if (c == NVS * 64 + 1 && db2) {
      atomicAdd(db2, s);
    }
  }
}

// dP -> dh rows (x collects the positive and the negative pair) and hit-embedding gradients
__global__ void __launch_bounds__(256) k_pairs_bwd(int64_t B, int d, int W, int hit_type, int n_hit_rows,
                                                   const float* __restrict__ dP, const int32_t* __restrict__ hit_idx,
                                                   float* __restrict__ dH, float* __restrict__ demb) {
  extern __shared__ float lacc[];  // [n_hit_rows, d] when the embedding is used
  const bool emb = (hit_type == TG_HIT_BIN || hit_type == TG_HIT_COUNT) && demb;
  if (emb) {
    for (int c = threadIdx.x; c < n_hit_rows * d; c += 256) lacc[c] = 0.f;
    __syncthreads();
  }
  const int lane = lane_id();
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < B; i += (int64_t)gridDim.x * 4) {
    const float* pp = dP + i * 2 * (int64_t)W;
    const float* pn = dP + (B + i) * 2 * (int64_t)W;
    int ipx = 0, ipy = 0, inx = 0, iny = 0;
    if (emb) {
      ipx = hit_idx[2 * i];
      ipy = hit_idx[2 * i + 1];
      inx = hit_idx[2 * (B + i)];
      iny = hit_idx[2 * (B + i) + 1];
    }
    for (int c = lane; c < d; c += TG_WAVE) {
      const float xp = pp[c], yp = pp[W + c], xn = pn[c], yn = pn[W + c];
      dH[i * d + c] = xp + xn;
      dH[(B + i) * d + c] = yp;
      dH[(2 * B + i) * d + c] = yn;
      if (emb) {
        atomicAdd(&lacc[ipx * d + c], xp);
        atomicAdd(&lacc[ipy * d + c], yp);
        atomicAdd(&lacc[inx * d + c], xn);
        atomicAdd(&lacc[iny * d + c], yn);
      }
    }
  }
  if (emb) {
    __syncthreads();
    for (int c = threadIdx.x; c < n_hit_rows * d; c += 256) atomicAdd(demb + c, lacc[c]);
  }
}

// ---------------------------------------------------------------------------------
// attention core, backward.  One wavefront per centre, two passes over its K key rows:
//   pass 1  p_hj = g_h . x_j,  da_hj = dS_h . x_j            (lane j keeps the pair)
//   softmax a_hj over the live keys;  ds_hj = a_hj (da_hj - sum_j a_hj da_hj)
//   pass 2  dG_h = sum_j ds_hj x_j;  dx_j = sum_h a_hj dS_h + ds_hj g_h
// dx_j's node part is added to the involved-node gradient row, its time part feeds the
// TimeEncode gradients (d cos(dt w + phi) = -sin(.) (dt dw + dphi)); edge features have no
// gradient.
// ---------------------------------------------------------------------------------
template <int NH, int NV>
__global__ void __launch_bounds__(256) k_attn_core_bwd(tg_model m, int64_t Q, const float* __restrict__ ts,
                                                       const int64_t* __restrict__ l1_nids,
                                                       const int64_t* __restrict__ l1_eids,
                                                       const float* __restrict__ l1_ts, const float4* __restrict__ reprs,
                                                       const uint64_t* __restrict__ bm, const uint32_t* __restrict__ rank,
                                                       const float4* __restrict__ G, const float4* __restrict__ dS,
                                                       float4* __restrict__ dG, float* __restrict__ dreprs,
                                                       float* __restrict__ tepart,
                                                       DropCfg dc, const float* __restrict__ dO,
                                                       const float* __restrict__ bv, const float4* __restrict__ key_rows,
                                                       float* __restrict__ dkey_rows) {
  // key_rows (the OUTER layer of --n_layers 2): the node part of key k of centre i is row i*K + k of a dense tensor - the
  // neighbour's embedding by the inner layer, no feature add - and its gradient goes to the same row of dkey_rows (plain
  // stores: every slot has one writer; slots of padding keys keep the zeros the caller put there)
  __shared__ float4 tred[4][2][NV][64];
  __shared__ float xp[4][NV * 256];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const uint64_t dkey = drop_key(dc);
  const int d = m.d, d4 = m.d / 4, e4 = m.d_e / 4, K = m.n_neighbors;
  const int kv4 = 2 * d4 + e4;
  const float4* nf = reinterpret_cast<const float4*>(m.nfeats);
  const float4* ef = reinterpret_cast<const float4*>(m.efeats);
  const float4* fq = reinterpret_cast<const float4*>(m.te_freq);
  const float4* ph = reinterpret_cast<const float4*>(m.te_phase);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 w4[NV], p4[NV], gw[NV], gp[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = lane + v * TG_WAVE;
    w4[v] = c < d4 ? fq[c] : z4;
    p4[v] = c < d4 ? ph[c] : z4;
    gw[v] = gp[v] = z4;
  }
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < Q; i += (int64_t)gridDim.x * 4) {
    int64_t nb_l = 0, eid_l = 0;
    float dt_l = 0.f;
    int u_l = 0;
    if (lane < K) {
      nb_l = l1_nids[i * K + lane];
      eid_l = l1_eids[i * K + lane];
      dt_l = ts[i] - l1_ts[i * K + lane];
      if (nb_l != 0) u_l = (int)bm_rank(bm, rank, nb_l);
    }
    const unsigned long long live0 = __ballot(nb_l != 0);
    float4 g[NH][3][NV], ds[NH][3][NV], dg[NH][3][NV];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float4* gh = G + ((int64_t)i * NH + h) * kv4;
      const float4* sh = dS + ((int64_t)i * NH + h) * kv4;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = lane + v * TG_WAVE;
        g[h][0][v] = c < d4 ? gh[c] : z4;
        g[h][1][v] = c < e4 ? gh[d4 + c] : z4;
        g[h][2][v] = c < d4 ? gh[d4 + e4 + c] : z4;
        ds[h][0][v] = c < d4 ? sh[c] : z4;
        ds[h][1][v] = c < e4 ? sh[d4 + c] : z4;
        ds[h][2][v] = c < d4 ? sh[d4 + e4 + c] : z4;
        dg[h][0][v] = dg[h][1][v] = dg[h][2][v] = z4;
      }
    }
    // raw rows of the next keys travel in a ring of PD register slots (see k_attn_core)
    constexpr int PD = NV == 1 ? 3 : 2;
    float4 ya[PD][NV], yn[PD][NV], yb[PD][NV];
    auto fetch = [&](int slot, int k) {
      const int64_t u = __shfl(u_l, k, TG_WAVE);
      const int64_t nb = __shfl(nb_l, k, TG_WAVE);
      const int64_t eid = __shfl(eid_l, k, TG_WAVE);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = lane + v * TG_WAVE;
        if (key_rows) {
          ya[slot][v] = c < d4 ? key_rows[(i * K + k) * d4 + c] : z4;
          yn[slot][v] = z4;
        } else {
          ya[slot][v] = c < d4 ? reprs[u * d4 + c] : z4;
          yn[slot][v] = (nf && c < d4) ? nf[nb * d4 + c] : z4;
        }
        yb[slot][v] = (ef && c < e4) ? ef[eid * e4 + c] : z4;
      }
    };
    auto build = [&](int slot, float dt, float4 (&x)[3][NV]) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = lane + v * TG_WAVE;
        float4 a = ya[slot][v];
        a.x += yn[slot][v].x; a.y += yn[slot][v].y; a.z += yn[slot][v].z; a.w += yn[slot][v].w;
        x[0][v] = a;
        x[1][v] = yb[slot][v];
        x[2][v] = c < d4 ? make_float4(time_enc(dt, w4[v].x, p4[v].x), time_enc(dt, w4[v].y, p4[v].y),
                                       time_enc(dt, w4[v].z, p4[v].z), time_enc(dt, w4[v].w, p4[v].w))
                         : z4;
      }
    };
    // ---- pass 1: scores and dS . x per key, kept by lane k
    float p_l[NH], da_l[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) p_l[h] = da_l[h] = 0.f;
    unsigned long long todo = live0;  // fetch cursor over the live keys, in list order
    auto next_key = [&]() {
      const int kk = todo ? (__ffsll(todo) - 1) : -1;
      todo &= todo - 1;
      return kk;
    };
    int ks[PD];
    auto prime = [&]() {
      todo = live0;
#pragma unroll
      for (int sl = 0; sl < PD; ++sl) {
        ks[sl] = next_key();
        if (ks[sl] >= 0) fetch(sl, ks[sl]);
      }
    };
    auto pass1 = [&](int slot, int k) {
      float4 x[3][NV];
      build(slot, __shfl(dt_l, k, TG_WAVE), x);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        float p = 0.f, q = 0.f;
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            p = dot4f(g[h][sgm][v], x[sgm][v], p);
            q = dot4f(ds[h][sgm][v], x[sgm][v], q);
          }
        p = wave_sum(p);
        q = wave_sum(q);
        if (lane == k) {
          p_l[h] = p;
          da_l[h] = q;
        }
      }
    };
    prime();
    while (ks[0] >= 0) {
      bool more = true;
#pragma unroll
      for (int sl = 0; sl < PD; ++sl) {
        if (more && ks[sl] >= 0) {
          pass1(sl, ks[sl]);
          ks[sl] = next_key();
          if (ks[sl] >= 0) fetch(sl, ks[sl]);
        } else {
          more = false;
        }
      }
    }
    // ---- softmax backward, one key per lane.  With attention dropout the weights that multiply
    // the values are a' = a * keep / (1 - p); the value bias enters as bv * sum_j a'_j, so
    // d a'_j also receives dr = dO_h . bv_h.
    const bool mine = (live0 >> lane) & 1ull;
    float a_l[NH], s_l[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float mx = wave_max(mine ? p_l[h] : -INFINITY);
      const float e = mine ? expf(p_l[h] - mx) : 0.f;
      const float l = wave_sum(e);
      const float a = live0 ? e / l : 0.f;
      float da = da_l[h], ms = 1.f;
      if (dc.p > 0.f) {
        const int dh = 2 * d / NH;
        float dr = 0.f;
        for (int c = lane; c < dh; c += TG_WAVE) dr = fmaf(dO[i * 2 * d + h * dh + c], bv[h * dh + c], dr);
        dr = wave_sum(dr);
        ms = (mine && drop_keep(dkey, DROP_ATTN, ((uint64_t)i * NH + h) * (uint64_t)K + (uint64_t)lane, dc.thresh))
                 ? dc.scale : 0.f;
        da = (da + dr) * ms;
      }
      const float dot = wave_sum(a * da);
      a_l[h] = a * ms;  // a' (what the values were weighted with)
      s_l[h] = a * (da - dot);
    }
    // ---- pass 2: dG and the key-row gradients
    auto pass2 = [&](int slot, int k) {
      const float dt = __shfl(dt_l, k, TG_WAVE);
      const int64_t u = __shfl(u_l, k, TG_WAVE);
      float4 x[3][NV];
      build(slot, dt, x);
      float4 dxn[NV], dxt[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) dxn[v] = dxt[v] = z4;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float a = __shfl(a_l[h], k, TG_WAVE), sk = __shfl(s_l[h], k, TG_WAVE);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          axpy4f(dg[h][0][v], sk, x[0][v]);
          axpy4f(dg[h][1][v], sk, x[1][v]);
          axpy4f(dg[h][2][v], sk, x[2][v]);
          axpy4f(dxn[v], a, ds[h][0][v]);
          axpy4f(dxn[v], sk, g[h][0][v]);
          axpy4f(dxt[v], a, ds[h][2][v]);
          axpy4f(dxt[v], sk, g[h][2][v]);
        }
      }
      // node part -> involved-node gradient row.  The row is transposed through LDS so that every
      // atomic instruction covers 64 CONSECUTIVE floats (two 128-byte lines) instead of 64 dwords
      // strided by 16 bytes (eight lines): the L2 atomic units work per cache line.
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = lane + v * TG_WAVE;
        if (c < d4) *reinterpret_cast<float4*>(&xp[wave][4 * c]) = dxn[v];
      }
      if (dkey_rows) {
        float* dr = dkey_rows + (i * K + k) * d;
        for (int c = lane; c < d; c += TG_WAVE) dr[c] = xp[wave][c];
      } else {
        float* dr = dreprs + u * d;
        for (int c = lane; c < d; c += TG_WAVE) atomicAdd(dr + c, xp[wave][c]);
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = lane + v * TG_WAVE;
        if (c < d4) {
          const float sx = -time_enc_sin(dt, w4[v].x, p4[v].x) * dxt[v].x;
          const float sy = -time_enc_sin(dt, w4[v].y, p4[v].y) * dxt[v].y;
          const float sz = -time_enc_sin(dt, w4[v].z, p4[v].z) * dxt[v].z;
          const float sw = -time_enc_sin(dt, w4[v].w, p4[v].w) * dxt[v].w;
          gp[v].x += sx; gp[v].y += sy; gp[v].z += sz; gp[v].w += sw;
          gw[v].x = fmaf(sx, dt, gw[v].x); gw[v].y = fmaf(sy, dt, gw[v].y);
          gw[v].z = fmaf(sz, dt, gw[v].z); gw[v].w = fmaf(sw, dt, gw[v].w);
        }
      }
    };
    prime();
    while (ks[0] >= 0) {
      bool more = true;
#pragma unroll
      for (int sl = 0; sl < PD; ++sl) {
        if (more && ks[sl] >= 0) {
          pass2(sl, ks[sl]);
          ks[sl] = next_key();
          if (ks[sl] >= 0) fetch(sl, ks[sl]);
        } else {
          more = false;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float4* oh = dG + ((int64_t)i * NH + h) * kv4;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = lane + v * TG_WAVE;
        if (c < d4) {
          oh[c] = dg[h][0][v];
          oh[d4 + e4 + c] = dg[h][2][v];
        }
        if (c < e4) oh[d4 + c] = dg[h][1][v];
      }
    }
  }
  // TimeEncode gradients: block reduction, then one atomic per column and block
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    tred[wave][0][v][lane] = gw[v];
    tred[wave][1][v][lane] = gp[v];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = lane + v * TG_WAVE;
      if (c < d4) {
        float4 s = tred[0][wave][v][lane];
        const float4 b = tred[1][wave][v][lane], e = tred[2][wave][v][lane], f = tred[3][wave][v][lane];
        s.x += b.x + e.x + f.x; s.y += b.y + e.y + f.y; s.z += b.z + e.z + f.z; s.w += b.w + e.w + f.w;
        // per-block partial row (plain store): [block][freq | phase][d]; k_te_reduce sums the blocks.
        // (768 blocks x 344 atomics on 344 addresses cost ~60 us here.)
        *reinterpret_cast<float4*>(tepart + ((int64_t)blockIdx.x * 2 + wave) * d + 4 * c) = s;
      }
    }
  }
}

// dfreq[c] += sum_b tepart[b][0][c], dphase[c] += sum_b tepart[b][1][c]
__global__ void __launch_bounds__(256) k_te_reduce(int nblocks, int d, const float* __restrict__ tepart,
                                                   float* __restrict__ dfreq, float* __restrict__ dphase) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;  // column of [freq | phase]
  // gridDim.y slices of the partial rows; one atomic per column and slice (16 per address)
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b_lo = blockIdx.y * per, b_hi = min(nblocks, b_lo + per);
  float s = 0.f;
  if (c < 2 * d)
    for (int b = b_lo + rg; b < b_hi; b += 4) s += tepart[(int64_t)b * 2 * d + c];
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < 2 * d) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(c < d ? dfreq + c : dphase + (c - d), t);
  }
}

// centre rows: dreprs[local(nid_i)] += dcc_i
__global__ void k_centre_scatter(int64_t Q, int d, const int64_t* __restrict__ nids, const uint64_t* __restrict__ bm,
                                 const uint32_t* __restrict__ rank, const float* __restrict__ dcc,
                                 float* __restrict__ dreprs) {
  const int64_t total = Q * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / d;
    atomicAdd(dreprs + (int64_t)bm_rank(bm, rank, nids[i]) * d + (t - i * d), dcc[t]);
  }
}

// constant half of the query projection: qconst[n] = bq[n] + sum_j Wq[n, d + j] cos(phi_j)
//   dWq[n, d + j] += dqconst[n] cos(phi_j);  dphi_j -= sin(phi_j) sum_n dqconst[n] Wq[n, d + j]
__global__ void __launch_bounds__(256) k_qconst_bwd(int d, const float* __restrict__ dqconst,
                                                    const float* __restrict__ wq, const float* __restrict__ freq,
                                                    const float* __restrict__ phase, float* __restrict__ dwq,
                                                    float* __restrict__ dbq, float* __restrict__ dphase) {
  // grid = (column blocks of 64, row slices): thread (c, rg) walks rows n = slice start + rg, +4, ...
  __shared__ float red[4][64];
  const int E = 2 * d;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int per = (E + gridDim.y - 1) / gridDim.y;
  const int n_lo = blockIdx.y * per, n_hi = min(E, n_lo + per);
  if (blockIdx.x == 0)
    for (int n = n_lo + threadIdx.x; n < n_hi; n += 256) dbq[n] += dqconst[n];
  float acc = 0.f, c = 0.f, s = 0.f;
  if (j < d) {
    c = time_enc(0.f, freq[j], phase[j]);
    s = time_enc_sin(0.f, freq[j], phase[j]);
    for (int n = n_lo + rg; n < n_hi; n += 4) {
      const float dq = dqconst[n];
      dwq[(int64_t)n * E + d + j] += dq * c;
      acc = fmaf(dq, wq[(int64_t)n * E + d + j], acc);
    }
  }
  red[rg][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rg == 0 && j < d) atomicAdd(dphase + j, -s * (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// ---------------------------------------------------------------------------------
// GRU cell backward (torch.nn.GRUCell): gate activations were kept by the forward epilogue
// ---------------------------------------------------------------------------------
__global__ void k_gru_bwd(tg_model m, int64_t cap, const int32_t* __restrict__ n_dev,
                          const int64_t* __restrict__ outdated, const int32_t* __restrict__ out_pos,
                          const float* __restrict__ gates, const float* __restrict__ dreprs, float* __restrict__ dgi,
                          float* __restrict__ dgh, int32_t* __restrict__ flags) {
  const int d = m.d;
  const int64_t n = min((int64_t)*n_dev, cap);
  if (blockIdx.x == 0 && threadIdx.x == 0 && flags) {
    flags[0] = 1;
    flags[1] = n > 0;
  }
  const float* hv = (m.upd_src == TG_SRC_LEFT) ? m.left_vals : m.right_vals;
  const int64_t total = n * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / d;
    const int j = (int)(t - i * d);
    const float* gp = gates + i * 4 * (int64_t)d + j;
    const float r = gp[0], z = gp[d], nn = gp[2 * d], hn = gp[3 * d];
    const float dh = dreprs[(int64_t)out_pos[i] * d + j];
    const float hold = hv[outdated[i] * d + j];
    const float dn = dh * (1.f - z) * (1.f - nn * nn);
    const float dz = dh * (hold - nn) * z * (1.f - z);
    const float dr = dn * hn * r * (1.f - r);
    float* gi = dgi + i * 3 * (int64_t)d + j;
    float* gh = dgh + i * 3 * (int64_t)d + j;
    gi[0] = dr; gi[d] = dz; gi[2 * d] = dn;
    gh[0] = dr; gh[d] = dz; gh[2 * d] = dn * r;
  }
}

// dh_new[i, :] = dreprs[out_pos[i], :] for the live outdated rows (MergeUpdater backward input)
__global__ void k_gather_dh(int64_t cap, const int32_t* __restrict__ n_dev, int d, const int32_t* __restrict__ out_pos,
                            const float* __restrict__ dreprs, float* __restrict__ dh, int32_t* __restrict__ flags) {
  const int64_t n = min((int64_t)*n_dev, cap);
  if (blockIdx.x == 0 && threadIdx.x == 0 && flags) {
    flags[0] = 1;
    flags[1] = n > 0;
  }
  const int64_t total = n * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / d;
    dh[t] = dreprs[(int64_t)out_pos[i] * d + (t - i * d)];
  }
}

// ---------------------------------------------------------------------------------
// Adam
// ---------------------------------------------------------------------------------
__global__ void k_adam_tick(int n_groups, const int32_t* __restrict__ enabled, int32_t* __restrict__ steps) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_groups && (!enabled || enabled[g])) steps[g] += 1;
}

__global__ void __launch_bounds__(256) k_adam(const tg_adam_seg* __restrict__ segs, const int32_t* __restrict__ enabled,
                                              const int32_t* __restrict__ steps, float lr, float b1, float b2, float eps,
                                              float gscale) {
  const tg_adam_seg sg = segs[blockIdx.y];
  if (enabled && !enabled[sg.group]) return;
  if ((int64_t)blockIdx.x * blockDim.x >= sg.n) return;
  __shared__ float sh[2];
  if (threadIdx.x == 0) {  // bias corrections in double, as the Python scalars of torch.optim.Adam
    const double t = (double)steps[sg.group];
    const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
    sh[0] = (float)((double)lr / bc1);
    sh[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float step_size = sh[0], bc2s = sh[1];
  const float gs = gscale * (sg.grad_scale != 0.f ? sg.grad_scale : 1.f);
  // four elements per thread in flight (the loop is a chain of memory round trips otherwise: 4.5 M parameters took 82 us)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < sg.n; i0 += 4 * stride) {
    float g[4], mo[4], vo[4], po[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = min(i0 + u * stride, sg.n - 1);
      g[u] = sg.g[i];
      mo[u] = sg.m[i];
      vo[u] = sg.v[i];
      po[u] = sg.p[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < sg.n) {
        const float gg = g[u] * gs;
        const float mm = mo[u] + (gg - mo[u]) * (1.f - b1);  // exp_avg.lerp_(grad, 1 - beta1)
        const float vv = vo[u] * b2 + (1.f - b2) * gg * gg;
        sg.m[i] = mm;
        sg.v[i] = vv;
        sg.p[i] = po[u] - step_size * (mm / (sqrtf(vv) / bc2s + eps));
      }
    }
  }
}

__global__ void k_rng_tick(uint64_t* rng) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1;
}

// ---------------------------------------------------------------------------------
// workspace of the training tail
// ---------------------------------------------------------------------------------
struct TrainWs {
  float *gates, *P, *T1, *dP, *dH, *dT, *dhh, *dcc, *dO, *dS, *dG, *dqp, *dreprs, *dgi, *dgh, *dqconst, *part, *tepart;
  float *dX, *dHn, *dt2, *dt0;  // non-default message transform / updater only
  size_t part_floats;
  int32_t* hit_idx;
  int64_t rows_cap;
  // hit windows when the step samples with another strategy than recent_edges (the windows are recent-edges lists always,
  // data_loader.py:61-66): lists of cat[src, dst, neg] + the sampler's other outputs (unused)
  int64_t *hit_nbr, *hit_eid;
  float* hit_ts;
  float* demb2;  // --n_layers 2: gradient of the neighbour slots' embeddings [Q*K, d]
};


static int score_width(const tg_model* m, const tg_score_params* sp) {
  return m->d + (sp->hit_type == TG_HIT_VEC ? m->n_neighbors : 0);
}

static size_t part_floats_for(const tg_model* m, const tg_score_params* sp) {
  // split partials of the grouped weight-gradient launch: at most 16 splits of every [N, K + 1] product
  // of the contrastive backward pass (also ample for the single launches of the restarter's backward)
  const size_t d = m->d, E = 2 * d, kvw = 2 * d + m->d_e, mw = 3 * d + m->d_e, W2 = 2 * (size_t)score_width(m, sp);
  const size_t total = d * (W2 + 1) + d * (d + 1) + d * (E + d + 1) + E * (E + 1) + 2 * E * (kvw + 1) + E * (d + 1) +
                       3 * d * (mw + 1) + 3 * d * (d + 1) + 2 * mw * (mw + 1) + d * (mw + d + 1) + d * (d + 1);
  return std::max<size_t>(16 * total + 64, (size_t)1280 * 4096);
}

// n_layers 2: the attention backward's temporaries serve both layers, one after the other, and are sized for the inner
// one (its centres are the Q*K neighbour slots)
static bool carve_train(const tg_model* m, const tg_score_params* sp, int64_t B, Carver& cv, TrainWs& w, int n_layers = 1) {
  const int64_t Q0 = 3 * B, K = m->n_neighbors;
  const int64_t Q = n_layers == 2 ? Q0 * K : Q0;  // rows of the attention backward's temporaries
  const int d = m->d, E = 2 * d, kvw = 2 * d + m->d_e, nh = m->n_head, W = score_width(m, sp);
  w.rows_cap = n_layers == 2 ? std::min<int64_t>(Q0 * (1 + K + K * K), std::max<int64_t>(m->n_nodes, 1))
                             : std::min<int64_t>(Q0 * (K + 1), m->n_nodes);
  w.gates = cv.take<float>((size_t)w.rows_cap * 4 * d);
  w.P = cv.take<float>((size_t)2 * B * 2 * W);
  w.T1 = cv.take<float>((size_t)2 * B * d);
  w.dP = cv.take<float>((size_t)2 * B * 2 * W);
  w.dH = cv.take<float>((size_t)Q0 * d);
  w.dT = cv.take<float>((size_t)Q * d);
  w.dhh = cv.take<float>((size_t)Q * E);
  w.dcc = cv.take<float>((size_t)Q * d);
  w.dO = cv.take<float>((size_t)Q * E);
  w.dS = cv.take<float>((size_t)Q * nh * kvw);
  w.dG = cv.take<float>((size_t)Q * nh * kvw);
  w.dqp = cv.take<float>((size_t)Q * E);
  w.dreprs = cv.take<float>((size_t)w.rows_cap * d);
  w.dgi = cv.take<float>((size_t)w.rows_cap * 3 * d);
  w.dgh = cv.take<float>((size_t)w.rows_cap * 3 * d);
  const size_t mw = 3 * (size_t)d + m->d_e;
  if (m->tsfm != TG_TSFM_ID) w.dX = cv.take<float>((size_t)w.rows_cap * mw);
  if (m->tsfm == TG_TSFM_MLP) w.dt0 = cv.take<float>((size_t)w.rows_cap * (mw / 2));
  if (m->upd_fn == TG_UPD_MERGE) {
    w.dHn = cv.take<float>((size_t)w.rows_cap * d);
    w.dt2 = cv.take<float>((size_t)w.rows_cap * d);
  }
  w.dqconst = cv.take<float>((size_t)E);
  w.tepart = cv.take<float>((size_t)1024 * 2 * d);
  w.part_floats = part_floats_for(m, sp);
  w.part = cv.take<float>(w.part_floats);
  w.hit_idx = cv.take<int32_t>((size_t)4 * B);
  w.hit_nbr = cv.take<int64_t>((size_t)Q0 * K);
  w.hit_eid = cv.take<int64_t>((size_t)Q0 * K);
  w.hit_ts = cv.take<float>((size_t)Q0 * K);
  w.demb2 = n_layers == 2 ? cv.take<float>((size_t)Q * d) : nullptr;
  return cv.ok;
}

static size_t train_ws_bytes(const tg_model* m, const tg_score_params* sp, int64_t B, int n_layers = 1) {
  char* const base = reinterpret_cast<char*>((uintptr_t)1 << 20);  // dry run of the carve above: never dereferenced
  Carver cv(base, (size_t)1 << 60);
  TrainWs w{};
  return carve_train(m, sp, B, cv, w, n_layers) ? (size_t)(cv.p - base) + 256 : 0;
}

static int train_supported(const tg_model* m, const tg_score_params* sp) {
  if (!attn_dims_ok(m) || !sp) return 0;
  if (m->tsfm == TG_TSFM_MLP && (((3 * m->d + m->d_e) / 2) % 4)) return 0;
  if (m->d > NVS * 64) return 0;
  if ((2 * score_width(m, sp)) % 4) return 0;
  if (sp->hit_type < TG_HIT_NONE || sp->hit_type > TG_HIT_COUNT) return 0;
  if ((sp->hit_type == TG_HIT_BIN || sp->hit_type == TG_HIT_COUNT) && (!sp->hit_emb || sp->n_hit_rows <= 0)) return 0;
  if (sp->hit_type == TG_HIT_COUNT && sp->n_hit_rows < m->n_neighbors + 1) return 0;
  if (sp->hit_type == TG_HIT_BIN && sp->n_hit_rows < 2) return 0;
  return 1;
}

// the recent-edges lists of cat[src, dst, neg] the hit windows are made of (data_loader.py:61-66: strategy='recent_edges'
// whatever the graph's own strategy): the step's own lists, or - another strategy - one more sampler launch over the
// step's query arrays (float64 times)
static const int64_t* hit_lists(const tg_tcsr* g, const tg_model* m, const tg_step_io* sio, StepWs& w, TrainWs& t, hipStream_t st,
                                int* rc) {
  *rc = TG_OK;
  if (sio->strategy == 0) return w.l1n;
  const int64_t Q = 3 * sio->B;
  *rc = tg_sample_recent_edges(g, Q, w.nids3, w.ts3, m->n_neighbors, t.hit_nbr, t.hit_eid, t.hit_ts, nullptr, nullptr, (void*)st);
  return t.hit_nbr;
}

// STEP 7 forward only (evaluation): scores and the BCE loss, no gradients
static int score_forward(const tg_model* m, const tg_tcsr* gr, const tg_train_io* io, StepWs& w, TrainWs& t, hipStream_t st) {
  const tg_score_params* sp = io->score;
  const int64_t B = io->step.B;
  const int d = m->d, K = m->n_neighbors;
  const int W2 = 2 * score_width(m, sp);
  int rc;
  const int64_t* hl = hit_lists(gr, m, &io->step, w, t, st, &rc);
  if (rc != TG_OK) return rc;
  hipLaunchKernelGGL(k_build_pairs, dim3(flat_grid(2 * B, 4)), dim3(256), 0, st, B, d, K, sp->hit_type, io->step.h,
                     w.nids3, hl, sp->hit_emb, t.P, t.hit_idx, io->losses);  // (zeroes the loss accumulators too)
  GemmArgs g{};
  g.m_cap = 2 * B; g.n = d; g.k = W2; g.a0 = ASeg{t.P, W2, W2, nullptr};
  g.w = sp->fc1.w; g.ldw = W2; g.bias = sp->fc1.b; g.c = t.T1; g.ldc = d; g.relu = 1; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  hipLaunchKernelGGL(k_score_loss, dim3(std::min<unsigned>(flat_grid(2 * B, 4), 256)), dim3(256), 0, st, B, d, t.T1,
                     sp->fc2.w, sp->fc2.b, io->pos_scores, io->neg_scores, io->losses, (float*)nullptr, (float*)nullptr,
                     DropCfg{});
  return check_launch("tg_train_step(scores)");
}

// Backward of ONE attention layer (temporal_agg_modules.py:186-235 + the merger, basic_modules.py:16-19) from the gradient
// of its output rows: weight-gradient products are appended to `tns` (the caller launches them grouped - they read the
// temporaries of this call, which the next layer's call re-uses), input-row gradients are added to t.dreprs (centres;
// neighbours unless the layer's keys were rows of a dense tensor: then into dkey_rows), TimeEncode gradients into the
// layer-independent gm->te_*.  dqconst: what k_qconst_bwd needs after the grouped launch.
struct AttnBwdIn {
  const tg_model *m, *gm;
  const AttnWs* a;
  int64_t Q;
  const int64_t* nids;
  const float* ts;
  const int64_t *l1n, *l1e;
  const float* l1t;
  const float* dH;        // [Q, d]
  const float* key_rows;  // nullable
  float* dkey_rows;       // nullable, with key_rows
};
static int attn_layer_backward(const AttnBwdIn& in, StepWs& w, TrainWs& t, const DropCfg& dc, std::vector<TnArgs>& tns,
                               hipStream_t st) {
  const tg_model* m = in.m;
  const tg_model* gm = in.gm;
  const AttnWs& a = *in.a;
  const int64_t Q = in.Q;
  const int d = m->d, d_e = m->d_e, E = 2 * d, kvw = 2 * d + d_e, nh = m->n_head, dh = E / nh;
  const float alpha = 1.0f / sqrtf((float)dh);
  auto F = [](const float* p) { return const_cast<float*>(p); };
  int rc;
  GemmArgs g{};
  TnArgs tn{};
  // ---- embedding merger (basic_modules.py:16-19): z = fc2(relu(fc1([hh | cc])))
  tn = TnArgs{};
  tn.m_cap = Q; tn.n = d; tn.k = d; tn.y = in.dH; tn.ldy = d; tn.x0 = ASeg{a.t, d, d, nullptr};
  tn.out = F(gm->attn_fc2.w); tn.ldo = d; tn.alpha = 1.f; tn.accumulate = 1; tn.nbatch = 1; tn.part = t.part;
  tn.part_floats = t.part_floats; tn.bias_out = F(gm->attn_fc2.b); tn.bias_accumulate = 1;
  tns.push_back(tn);
  g = GemmArgs{};
  g.m_cap = Q; g.n = d; g.k = d; g.a0 = ASeg{in.dH, d, d, nullptr};
  g.w = m->attn_fc2.w; g.ldw = d; g.w_kmajor = 1; g.c = t.dT; g.ldc = d; g.alpha = 1.f; g.nbatch = 1;
  g.relu_mask = a.t; g.ld_mask = d;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  tn = TnArgs{};
  tn.m_cap = Q; tn.n = d; tn.k = E + d; tn.y = t.dT; tn.ldy = d;
  tn.x0 = ASeg{a.hh, E, E, nullptr}; tn.x1 = ASeg{a.cc, d, d, nullptr};
  tn.out = F(gm->attn_fc1.w); tn.ldo = E + d; tn.alpha = 1.f; tn.accumulate = 1; tn.nbatch = 1; tn.part = t.part;
  tn.part_floats = t.part_floats; tn.bias_out = F(gm->attn_fc1.b); tn.bias_accumulate = 1;
  tns.push_back(tn);
  // d hh (zero for centres without neighbours: their hh was masked) and the direct part of d cc
  g = GemmArgs{};
  g.m_cap = Q; g.n = E; g.k = d; g.a0 = ASeg{t.dT, d, d, nullptr};
  g.w = m->attn_fc1.w; g.ldw = E + d; g.w_kmajor = 1; g.c = t.dhh; g.ldc = E; g.alpha = 1.f; g.nbatch = 1;
  g.row_valid = a.valid;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  g = GemmArgs{};
  g.m_cap = Q; g.n = d; g.k = d; g.a0 = ASeg{t.dT, d, d, nullptr};
  g.w = m->attn_fc1.w + E; g.ldw = E + d; g.w_kmajor = 1; g.c = t.dcc; g.ldc = d; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // ---- out projection
  tn = TnArgs{};
  tn.m_cap = Q; tn.n = E; tn.k = E; tn.y = t.dhh; tn.ldy = E; tn.x0 = ASeg{a.o, E, E, nullptr};
  tn.out = F(gm->attn_out.w); tn.ldo = E; tn.alpha = 1.f; tn.accumulate = 1; tn.nbatch = 1; tn.part = t.part;
  tn.part_floats = t.part_floats; tn.bias_out = F(gm->attn_out.b); tn.bias_accumulate = 1;
  tns.push_back(tn);
  g = GemmArgs{};
  g.m_cap = Q; g.n = E; g.k = E; g.a0 = ASeg{t.dhh, E, E, nullptr};
  g.w = m->attn_out.w; g.ldw = E; g.w_kmajor = 1; g.c = t.dO; g.ldc = E; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // ---- value projection: o_h = Wv_h s_h + bv_h
  tn = TnArgs{};
  tn.m_cap = Q; tn.n = dh; tn.k = kvw; tn.y = t.dO; tn.ldy = E; tn.y_bs = dh;
  tn.x0 = ASeg{a.s, (int64_t)nh * kvw, kvw, nullptr}; tn.x0_bs = kvw;
  tn.out = F(gm->attn_wv); tn.ldo = kvw; tn.out_bs = (int64_t)dh * kvw; tn.alpha = 1.f; tn.accumulate = 1; tn.nbatch = nh;
  tn.part = t.part; tn.part_floats = t.part_floats;
  tn.bias_out = F(gm->attn_b_in) + 2 * E; tn.bias_accumulate = 1; tn.bias_bs = dh;
  if (dc.p > 0.f) { tn.bias_rs = a.rsum; tn.ld_brs = nh; tn.brs_col = 0; }  // d bv_h = sum_i r_ih dO_ih
  tns.push_back(tn);
  g = GemmArgs{};
  g.m_cap = Q; g.n = kvw; g.k = dh; g.a0 = ASeg{t.dO, E, dh, nullptr}; g.a0_bs = dh;
  g.w = m->attn_wv; g.ldw = kvw; g.w_kmajor = 1; g.w_bs = (int64_t)dh * kvw;
  g.c = t.dS; g.ldc = (int64_t)nh * kvw; g.c_bs = kvw; g.alpha = 1.f; g.nbatch = nh;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // ---- softmax / gather core
  const int nv = (int)cdiv(std::max(d, d_e) / 4, TG_WAVE);
  const unsigned cgrid = std::min<unsigned>(flat_grid(Q, 4), 1024);
#define TG_CORE_BWD(NH_, NV_)                                                                                       \
  hipLaunchKernelGGL((k_attn_core_bwd<NH_, NV_>), dim3(cgrid), dim3(256), 0, st, *m, Q, in.ts, in.l1n, in.l1e, in.l1t, \
                     (const float4*)w.reprs, w.bm, w.rank, (const float4*)a.g, (const float4*)t.dS, (float4*)t.dG,    \
                     t.dreprs, t.tepart, dc, t.dO, m->attn_b_in + 2 * E, (const float4*)in.key_rows, in.dkey_rows)
  if (nh == 2 && nv == 1) TG_CORE_BWD(2, 1);
  else if (nh == 2 && nv == 2) TG_CORE_BWD(2, 2);
  else if (nh == 1 && nv == 1) TG_CORE_BWD(1, 1);
  else if (nh == 4 && nv == 1) TG_CORE_BWD(4, 1);
  else return TG_EUNSUPPORTED;
#undef TG_CORE_BWD
  hipLaunchKernelGGL(k_te_reduce, dim3((unsigned)cdiv(2 * d, 64), 16), dim3(256), 0, st, (int)cgrid, d, t.tepart,
                     F(gm->te_freq), F(gm->te_phase));
  // ---- key projection folded into the query: g_h = Wk_h^T q_h
  tn = TnArgs{};
  tn.m_cap = Q; tn.n = dh; tn.k = kvw; tn.y = a.qp; tn.ldy = E; tn.y_bs = dh;
  tn.x0 = ASeg{t.dG, (int64_t)nh * kvw, kvw, nullptr}; tn.x0_bs = kvw;
  tn.out = F(gm->attn_wk); tn.ldo = kvw; tn.out_bs = (int64_t)dh * kvw; tn.alpha = 1.f; tn.accumulate = 1; tn.nbatch = nh;
  tn.part = t.part; tn.part_floats = t.part_floats;
  tns.push_back(tn);
  g = GemmArgs{};
  g.m_cap = Q; g.n = dh; g.k = kvw; g.a0 = ASeg{t.dG, (int64_t)nh * kvw, kvw, nullptr}; g.a0_bs = kvw;
  g.w = m->attn_wk; g.ldw = kvw; g.w_bs = (int64_t)dh * kvw;
  g.c = t.dqp; g.ldc = E; g.c_bs = dh; g.alpha = 1.f; g.nbatch = nh;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // ---- query projection: qp = alpha (Wq[:, :d] cc + qconst)
  tn = TnArgs{};
  tn.m_cap = Q; tn.n = E; tn.k = d; tn.y = t.dqp; tn.ldy = E; tn.x0 = ASeg{a.cc, d, d, nullptr};
  tn.out = F(gm->attn_wq); tn.ldo = E; tn.alpha = alpha; tn.accumulate = 1; tn.nbatch = 1; tn.part = t.part;
  tn.part_floats = t.part_floats; tn.bias_out = t.dqconst; tn.bias_accumulate = 0;
  tns.push_back(tn);
  g = GemmArgs{};
  g.m_cap = Q; g.n = d; g.k = E; g.a0 = ASeg{t.dqp, E, E, nullptr};
  g.w = m->attn_wq; g.ldw = E; g.w_kmajor = 1; g.c = t.dcc; g.ldc = d; g.alpha = alpha; g.nbatch = 1; g.accumulate = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  hipLaunchKernelGGL(k_centre_scatter, dim3(flat_grid(Q * d, 256)), dim3(256), 0, st, Q, d, in.nids, w.bm, w.rank, t.dcc,
                     t.dreprs);
  return TG_OK;
}

// backward of the contrastive loss; everything it reads is still in the step workspace
static int contrast_backward(const tg_model* m, const tg_tcsr* gr, const tg_train_io* io, StepWs& w, TrainWs& t,
                             const DropCfg& dc, hipStream_t st) {
  const tg_score_params* sp = io->score;
  const tg_model* gm = io->grads;
  const tg_score_params* gs = io->score_grads;
  const int64_t B = io->step.B, Q = 3 * B;
  const int d = m->d, d_e = m->d_e, K = m->n_neighbors;
  const int W = score_width(m, sp), W2 = 2 * W;
  const int mw = 3 * d + d_e;
  int rc;
  std::vector<TnArgs> tns;  // weight-gradient products, launched together at the end
  hipError_t e = hipMemsetAsync(t.dreprs, 0, (size_t)t.rows_cap * d * sizeof(float), st);
  if (e == hipSuccess) e = hipMemsetAsync(io->losses, 0, 2 * sizeof(float), st);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_train_step memset");
    return TG_EHIP;
  }
  auto F = [](const float* p) { return const_cast<float*>(p); };
  // ---- STEP 7 forward
  const int64_t* hl = hit_lists(gr, m, &io->step, w, t, st, &rc);
  if (rc != TG_OK) return rc;
  hipLaunchKernelGGL(k_build_pairs, dim3(flat_grid(2 * B, 4)), dim3(256), 0, st, B, d, K, sp->hit_type, io->step.h,
                     w.nids3, hl, sp->hit_emb, t.P, t.hit_idx, (float*)nullptr);
  GemmArgs g{};
  g.m_cap = 2 * B; g.n = d; g.k = W2; g.a0 = ASeg{t.P, W2, W2, nullptr};
  g.w = sp->fc1.w; g.ldw = W2; g.bias = sp->fc1.b; g.c = t.T1; g.ldc = d; g.relu = 1; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // fc1 input gradient needs relu(T1) as the mask, and k_score_loss overwrites T1 with dT1: since
  // dT1 is zero exactly where T1 <= 0, the masked gradient IS dT1 and no copy of T1 is needed
  hipLaunchKernelGGL(k_score_loss, dim3(std::min<unsigned>(flat_grid(2 * B, 4), 256)), dim3(256), 0, st, B, d, t.T1,
                     sp->fc2.w, sp->fc2.b, io->pos_scores, io->neg_scores, io->losses, F(gs->fc2.w), F(gs->fc2.b), dc);
  // ---- STEP 7 backward: score MergeLayer
  TnArgs tn{};
  tn.m_cap = 2 * B; tn.n = d; tn.k = W2; tn.y = t.T1; tn.ldy = d; tn.x0 = ASeg{t.P, W2, W2, nullptr};
  tn.out = F(gs->fc1.w); tn.ldo = W2; tn.alpha = 1.f; tn.accumulate = 1; tn.nbatch = 1; tn.part = t.part;
  tn.part_floats = t.part_floats; tn.bias_out = F(gs->fc1.b); tn.bias_accumulate = 1;
  tns.push_back(tn);
  g = GemmArgs{};
  g.m_cap = 2 * B; g.n = W2; g.k = d; g.a0 = ASeg{t.T1, d, d, nullptr};
  g.w = sp->fc1.w; g.ldw = W2; g.w_kmajor = 1; g.c = t.dP; g.ldc = W2; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  const bool emb = sp->hit_type == TG_HIT_BIN || sp->hit_type == TG_HIT_COUNT;
  const size_t lbytes = emb ? (size_t)sp->n_hit_rows * d * sizeof(float) : 0;
  if (lbytes > 60 * 1024) return TG_EUNSUPPORTED;
  hipLaunchKernelGGL(k_pairs_bwd, dim3(std::min<unsigned>(flat_grid(B, 4), 256)), dim3(256), lbytes, st, B, d, W,
                     sp->hit_type, sp->n_hit_rows, t.dP, t.hit_idx, t.dH, emb ? F(gs->hit_emb) : (float*)nullptr);
  // ---- the attention layer(s).  --n_layers 2: the outer layer (fns[0]) over the batch's Q centres with the neighbour
  // slots' embeddings as key rows, then the inner layer (fns[1]) over the Q*K slots at the roots' query times
  // (temporal_agg_modules.py:57-66); the outer layer's weight-gradient products are launched before the inner layer
  // re-uses the temporaries they read
  const tg_model* inner = w.h2n ? io->step.inner : nullptr;
  AttnBwdIn ab{m, gm, &w.attn, Q, w.nids3, w.ts3f, w.l1n, w.l1e, w.l1t, t.dH, nullptr, nullptr};
  if (inner) {
    if (!io->inner_grads || !t.demb2) return TG_EINVAL;
    if ((e = hipMemsetAsync(t.demb2, 0, (size_t)Q * K * d * sizeof(float), st)) != hipSuccess) return TG_EHIP;
    ab.key_rows = w.emb2;
    ab.dkey_rows = t.demb2;
  }
  if ((rc = attn_layer_backward(ab, w, t, dc, tns, st)) != TG_OK) return rc;
  if (inner) {
    if ((rc = gemm_tn_group_launch(tns.data(), (int)tns.size(), t.part, t.part_floats, st)) != TG_OK) return rc;
    tns.clear();
    hipLaunchKernelGGL(k_qconst_bwd, dim3((unsigned)cdiv(d, 64), 16), dim3(256), 0, st, d, t.dqconst, m->attn_wq, m->te_freq,
                       m->te_phase, F(gm->attn_wq), F(gm->attn_b_in), F(gm->te_phase));
    const AttnBwdIn ab2{inner, io->inner_grads, &w.attn2, Q * K, w.l1n, w.ts2, w.h2n, w.h2e, w.h2t, t.demb2, nullptr, nullptr};
    if ((rc = attn_layer_backward(ab2, w, t, dc, tns, st)) != TG_OK) return rc;
  }
  const tg_model* m_last = inner ? inner : m;        // the layer whose dqconst is pending
  const tg_model* gm_last = inner ? io->inner_grads : gm;
  // ---- updater (update_modules.py:30-47) and message transform (message_modules.py:20-55)
  const float* upd_vals = (m->upd_src == TG_SRC_LEFT) ? m->left_vals : m->right_vals;
  const int32_t* n_out = w.counts + 1;
  // the forward pass left the transformed messages / hidden layers in its own scratch (apply_messages)
  Carver av(w.apply_ws, w.apply_bytes);
  const float* t0 = m->tsfm == TG_TSFM_MLP ? av.take<float>((size_t)(Q * (K + 1)) * (mw / 2)) : nullptr;
  const float* t1 = m->tsfm != TG_TSFM_ID ? av.take<float>((size_t)(Q * (K + 1)) * mw) : nullptr;
  const float* t2 = m->upd_fn == TG_UPD_MERGE ? av.take<float>((size_t)(Q * (K + 1)) * d) : nullptr;
  const ASeg xmsg = m->tsfm == TG_TSFM_ID ? ASeg{m->msg_vals, mw, mw, w.outdated} : ASeg{t1, mw, mw, nullptr};
  const ASeg xraw = ASeg{m->msg_vals, mw, mw, w.outdated};
  const ASeg xmem = ASeg{upd_vals, d, d, w.outdated};
  auto tn_rows = [&]() {
    TnArgs a{};
    a.m_cap = t.rows_cap; a.m_dev = n_out; a.alpha = 1.f; a.accumulate = 1; a.nbatch = 1; a.bias_accumulate = 1;
    return a;
  };
  const bool need_dx = m->tsfm != TG_TSFM_ID;
  if (m->upd_fn == TG_UPD_GRU) {
    hipLaunchKernelGGL(k_gru_bwd, dim3(flat_grid(t.rows_cap * d, 256)), dim3(256), 0, st, *m, t.rows_cap, n_out,
                       w.outdated, w.out_pos, t.gates, t.dreprs, t.dgi, t.dgh, io->flags);
    tn = tn_rows();
    tn.n = 3 * d; tn.k = mw; tn.y = t.dgi; tn.ldy = 3 * d; tn.x0 = xmsg;
    tn.out = F(gm->gru_w_ih); tn.ldo = mw; tn.bias_out = F(gm->gru_b_ih);
    tns.push_back(tn);
    tn.k = d; tn.y = t.dgh; tn.x0 = xmem; tn.out = F(gm->gru_w_hh); tn.ldo = d; tn.bias_out = F(gm->gru_b_hh);
    tns.push_back(tn);
    if (need_dx) {  // d msg = d gi W_ih
      g = GemmArgs{};
      g.m_cap = t.rows_cap; g.m_dev = n_out; g.n = mw; g.k = 3 * d; g.a0 = ASeg{t.dgi, 3 * d, 3 * d, nullptr};
      g.w = m->gru_w_ih; g.ldw = mw; g.w_kmajor = 1; g.c = t.dX; g.ldc = mw; g.alpha = 1.f; g.nbatch = 1;
      if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
    }
  } else {  // MergeUpdater: h = fc2(relu(fc1([msg | mem])))
    hipLaunchKernelGGL(k_gather_dh, dim3(flat_grid(t.rows_cap * d, 256)), dim3(256), 0, st, t.rows_cap, n_out, d,
                       w.out_pos, t.dreprs, t.dHn, io->flags);
    tn = tn_rows();
    tn.n = d; tn.k = d; tn.y = t.dHn; tn.ldy = d; tn.x0 = ASeg{t2, d, d, nullptr};
    tn.out = F(gm->upd_fc2.w); tn.ldo = d; tn.bias_out = F(gm->upd_fc2.b);
    tns.push_back(tn);
    g = GemmArgs{};
    g.m_cap = t.rows_cap; g.m_dev = n_out; g.n = d; g.k = d; g.a0 = ASeg{t.dHn, d, d, nullptr};
    g.w = m->upd_fc2.w; g.ldw = d; g.w_kmajor = 1; g.c = t.dt2; g.ldc = d; g.alpha = 1.f; g.nbatch = 1;
    g.relu_mask = t2; g.ld_mask = d;
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
    tn = tn_rows();
    tn.n = d; tn.k = mw + d; tn.y = t.dt2; tn.ldy = d; tn.x0 = xmsg; tn.x1 = xmem;
    tn.out = F(gm->upd_fc1.w); tn.ldo = mw + d; tn.bias_out = F(gm->upd_fc1.b);
    tns.push_back(tn);
    if (need_dx) {  // d msg = d t2 fc1[:, :mw]
      g = GemmArgs{};
      g.m_cap = t.rows_cap; g.m_dev = n_out; g.n = mw; g.k = d; g.a0 = ASeg{t.dt2, d, d, nullptr};
      g.w = m->upd_fc1.w; g.ldw = mw + d; g.w_kmajor = 1; g.c = t.dX; g.ldc = mw; g.alpha = 1.f; g.nbatch = 1;
      if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
    }
  }
  if (m->tsfm == TG_TSFM_LINEAR) {  // msg = W raw + b  (raw messages carry no gradient: tgn_mode, tiger.py:329-333)
    tn = tn_rows();
    tn.n = mw; tn.k = mw; tn.y = t.dX; tn.ldy = mw; tn.x0 = xraw;
    tn.out = F(gm->tsfm1.w); tn.ldo = mw; tn.bias_out = F(gm->tsfm1.b);
    tns.push_back(tn);
  } else if (m->tsfm == TG_TSFM_MLP) {  // msg = W2 relu(W1 raw + b1) + b2
    const int hw = mw / 2;
    tn = tn_rows();
    tn.n = mw; tn.k = hw; tn.y = t.dX; tn.ldy = mw; tn.x0 = ASeg{t0, hw, hw, nullptr};
    tn.out = F(gm->tsfm2.w); tn.ldo = hw; tn.bias_out = F(gm->tsfm2.b);
    tns.push_back(tn);
    g = GemmArgs{};
    g.m_cap = t.rows_cap; g.m_dev = n_out; g.n = hw; g.k = mw; g.a0 = ASeg{t.dX, mw, mw, nullptr};
    g.w = m->tsfm2.w; g.ldw = hw; g.w_kmajor = 1; g.c = t.dt0; g.ldc = hw; g.alpha = 1.f; g.nbatch = 1;
    g.relu_mask = t0; g.ld_mask = hw;
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
    tn = tn_rows();
    tn.n = hw; tn.k = mw; tn.y = t.dt0; tn.ldy = hw; tn.x0 = xraw;
    tn.out = F(gm->tsfm1.w); tn.ldo = mw; tn.bias_out = F(gm->tsfm1.b);
    tns.push_back(tn);
  }
  if ((rc = gemm_tn_group_launch(tns.data(), (int)tns.size(), t.part, t.part_floats, st)) != TG_OK) return rc;
  hipLaunchKernelGGL(k_qconst_bwd, dim3((unsigned)cdiv(d, 64), 16), dim3(256), 0, st, d, t.dqconst, m_last->attn_wq, m->te_freq,
                     m->te_phase, F(gm_last->attn_wq), F(gm_last->attn_b_in), F(gm->te_phase));
  return check_launch("tg_train_step(backward)");
}

}  // namespace tg

namespace tg {
// A second stream for the mutual half's forward (tg_restart.hip: mutual_step phase 1): the restarter's forward reads the batch,
// the graph and its own parameters only, the contrast half (STEP 7, backward, STEP 4-5) nothing of the restarter's - two
// chains of short, latency-bound launches that fill the chip together.  fork / join by events: under stream capture the
// lane becomes a parallel branch of the graph.  Created by the first call outside a capture; TG_TRAIN_SIDE=0 keeps one stream.
struct TrainLane {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  hipEvent_t pool[16] = {};  // mutual_step's weight-gradient launches (tg_step.h: SideCtx)
  bool ok = false;
};
static TrainLane* train_lane(hipStream_t st) {
  static const int knob = getenv("TG_TRAIN_SIDE") ? atoi(getenv("TG_TRAIN_SIDE")) : 1;  // tuning knob
  if (!knob) return nullptr;
  static TrainLane lanes[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  TrainLane& L = lanes[dev];
  if (L.ok) return &L;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;  // not now: no stream / event is created inside a capture
  }
  bool good = hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&L.fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&L.join, hipEventDisableTiming) == hipSuccess;
  for (int j = 0; j < 16 && good; ++j) good = hipEventCreateWithFlags(&L.pool[j], hipEventDisableTiming) == hipSuccess;
  if (!good) {
    (void)hipGetLastError();
    return nullptr;
  }
  L.ok = true;
  return &L;
}
}  // namespace tg

using namespace tg;

extern "C" size_t tg_train_step_workspace_bytes(const tg_model* m, const tg_score_params* sp, int32_t restarter,
                                                const tg_seq_restarter* seq, int64_t B) {
  return tg_train_step_workspace_bytes2(m, sp, restarter, seq, B, 1);
}
extern "C" size_t tg_train_step_workspace_bytes2(const tg_model* m, const tg_score_params* sp, int32_t restarter,
                                                 const tg_seq_restarter* seq, int64_t B, int32_t n_layers) {
  if (!m || !train_supported(m, sp) || B <= 0 || m->n_nodes <= 0 || (n_layers != 1 && n_layers != 2)) return 0;
  if (restarter < TG_RESTARTER_NONE || restarter > TG_RESTARTER_STATIC || (restarter == TG_RESTARTER_SEQ && !seq)) return 0;
  size_t b = tg_stream_step_workspace_bytes2(m, B, n_layers) + train_ws_bytes(m, sp, B, n_layers);
  if (restarter != TG_RESTARTER_NONE) b += mutual_ws_bytes(m, restarter == TG_RESTARTER_SEQ ? seq : nullptr, B);
  return b;
}

extern "C" int tg_train_step(const tg_model* m_in, const tg_tcsr* g, const tg_train_io* io, void* ws,
                             size_t ws_bytes, void* stream) {
  if (m_in && m_in->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!m_in || !io) return TG_EINVAL;
  const bool eval_only = !io->grads;  // evaluation: forward + STEP 7 scores + write-back, nothing else
  tg_model m_local = *m_in;
  if (!eval_only) m_local.attn_fused = nullptr;  // the backward pass is built on the unfused forward's intermediates
  const tg_model* m = &m_local;
  if (!g || !io->score || (!eval_only && !io->score_grads) || !io->losses) return TG_EINVAL;
  if (!train_supported(m, io->score)) return TG_EUNSUPPORTED;
  const tg_step_io* sio = &io->step;
  if (sio->B <= 0 || !sio->src || !sio->dst || !sio->neg || !sio->ts || !sio->eids || !sio->h || !sio->err) return TG_EINVAL;
  if (sio->embed_only || g->num_node != m->n_nodes) return TG_EINVAL;
  if (io->restarter < TG_RESTARTER_NONE || io->restarter > TG_RESTARTER_STATIC) return TG_EINVAL;
  hipStream_t st = as_stream(stream);
  Carver cv(ws, ws_bytes);
  StepWs w{};
  TrainWs t{};
  const int n_layers = sio->inner ? 2 : 1;
  tg_model inner_local{};
  tg_step_io sio_local = *sio;
  if (sio->inner) {  // the second layer's weights; as for the first, the backward pass needs the unfused forward
    inner_local = *sio->inner;
    if (!eval_only) inner_local.attn_fused = nullptr;
    sio_local.inner = &inner_local;
    sio = &sio_local;
    if (!eval_only && !io->inner_grads) return TG_EINVAL;
  }
  if (!carve_step(m, sio->B, cv, w, n_layers) || !carve_train(m, io->score, sio->B, cv, t, n_layers)) return TG_EWORKSPACE;
  int rc;
  if (eval_only) {
    // a model that streams with eager updates (tg_model.pending_vals): the evaluation step is the streaming step's
    // forward in whatever form tg_stream_step would take for this io (table-backed, lean, riders ...) + STEP 7
    const bool eager = m->pending_vals != nullptr;
    if ((rc = step_forward(m, g, sio, w, nullptr, st, nullptr, nullptr, eager)) != TG_OK) return rc;
    if ((rc = score_forward(m, g, io, w, t, st)) != TG_OK) return rc;
    if ((rc = step_writeback_a(m, sio, w, st, nullptr)) != TG_OK) return rc;
    return step_writeback_b(m, g, sio, w, st, nullptr);
  }
  if (io->dropout_p < 0.f || io->dropout_p >= 1.f || (io->dropout_p > 0.f && !io->rng)) return TG_EINVAL;
  const DropCfg dc = make_drop(io->dropout_p, io->rng);
  const bool seq = io->restarter == TG_RESTARTER_SEQ;
  if (seq && (!io->seq || !io->seq_grads)) return TG_EINVAL;
  // the SeqRestarter's forward runs beside the contrast half on a second stream.  It needs the batch's id list and nothing
  // else of the forward pass: the fork is recorded INSIDE step_forward, right behind the sampler's launch, so the lane's
  // 0.6 ms overlap the forward pass as well as the backward (TG_TRAIN_FORK=0: fork behind the whole forward pass)
  static const int fork_knob = getenv("TG_TRAIN_FORK") ? atoi(getenv("TG_TRAIN_FORK")) : 1;  // tuning knob
  TrainLane* lane = seq ? train_lane(st) : nullptr;
  if (lane && fork_knob) w.collate_done = lane->fork;
  if ((rc = step_forward(m, g, sio, w, t.gates, st, nullptr, &dc)) != TG_OK) return rc;
  SideCtx side{};
  auto mutual = [&](int phase, hipStream_t s, SideCtx* sd) {
    return mutual_step(m, g, sio, w, seq ? io->seq : nullptr, seq ? io->seq_grads : nullptr, io->static_left,
                       io->static_right, io->static_left_grad, io->static_right_grad, io->losses + 1,
                       io->flags ? io->flags + 2 : nullptr, t.part, t.part_floats, cv.p, cv.left, dc, s, phase, sd);
  };
  if (lane && !((w.collate_recorded || hipEventRecord(lane->fork, st) == hipSuccess) &&
                hipStreamWaitEvent(lane->s, lane->fork, 0) == hipSuccess)) {
    (void)hipGetLastError();
    lane = nullptr;
  }
  if (lane && (rc = mutual(1, lane->s, nullptr)) != TG_OK) return rc;
  if ((rc = contrast_backward(m, g, io, w, t, dc, st)) != TG_OK) return rc;
  if ((rc = step_writeback_a(m, sio, w, st, nullptr)) != TG_OK) return rc;
  if (lane && !(hipEventRecord(lane->join, lane->s) == hipSuccess && hipStreamWaitEvent(st, lane->join, 0) == hipSuccess)) {
    set_hip_error(hipGetLastError(), "tg_train_step lane join");
    return TG_EHIP;
  }
  if (io->restarter != TG_RESTARTER_NONE) {  // needs the targets of STEP 4/5 and the step's bitmap-free inputs
    if (lane) {
      side.s = lane->s;
      side.n = 16;
      for (int j = 0; j < 16; ++j) side.ev[j] = lane->pool[j];
    }
    if ((rc = mutual(lane ? 2 : 3, st, lane ? &side : nullptr)) != TG_OK) return rc;
  } else if (io->flags) {
    (void)hipMemsetAsync(io->flags + 2, 0, sizeof(int32_t), st);
  }
  if (dc.p > 0.f) hipLaunchKernelGGL(k_rng_tick, dim3(1), dim3(64), 0, st, io->rng);
  return step_writeback_b(m, g, sio, w, st, nullptr);
}

// ---- restart-mode evaluation over consecutive batches as one call (tiger_hip.h: tg_restart_run) -------------------------
namespace tg {
__global__ void k_inc_i64(int64_t* p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1;
}
struct EvalLane {
  hipStream_t s = nullptr;
  hipEvent_t counted[TG_RUN_CTX] = {}, fwd_done[2] = {}, applied[2] = {}, join = nullptr;
  bool ok = false;
};
static EvalLane* eval_lane() {
  static EvalLane lanes[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  EvalLane& L = lanes[dev];
  if (L.ok) return &L;
  bool good = hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&L.join, hipEventDisableTiming) == hipSuccess;
  for (int j = 0; j < TG_RUN_CTX && good; ++j) good = hipEventCreateWithFlags(&L.counted[j], hipEventDisableTiming) == hipSuccess;
  for (int j = 0; j < 2 && good; ++j)
    good = hipEventCreateWithFlags(&L.fwd_done[j], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&L.applied[j], hipEventDisableTiming) == hipSuccess;
  if (!good) {
    (void)hipGetLastError();
    return nullptr;
  }
  L.ok = true;
  return &L;
}
}  // namespace tg

extern "C" int tg_eval_restart_run(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, const tg_train_io* step_io,
                                   void* step_ws, size_t step_ws_bytes, const tg_restart_run* run, int64_t count,
                                   void* stream) {
  if (!m || !g || !step_io || !run || count < 0) return TG_EINVAL;
  if (count == 0) return TG_OK;
  const int G = run->group;
  if (G < 1 || G > TG_RESTART_MAX_LISTS || step_io->grads || !run->g_restart || !run->offsets || run->cap <= 0 ||
      run->rows_cap <= 0 || run->fwd_nodes <= 0 || (r ? !run->fwd_ws : (!run->static_left || !run->static_right)))
    return TG_EINVAL;
  for (int j = 0; j < 2 * G; ++j) {
    const tg_step_io* p = run->pass_io[j];
    if (!p || !p->collate_only || !p->lazy || !p->lazy->list || !p->lazy->tmin || !p->lazy->keep_msg_bits || !p->counts ||
        !run->pass_ws[j] || !run->count_host[j])
      return TG_EINVAL;
  }
  for (int j = 0; j < 2; ++j)
    if (!run->ids[j] || !run->h_left[j] || !run->h_right[j] || !run->prev_ts[j]) return TG_EINVAL;
  hipStream_t st = as_stream(stream);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return TG_EUNSUPPORTED;  // the host reads a count per batch
  }
  EvalLane* L = eval_lane();
  if (!L) return TG_EHIP;
  const int64_t B = step_io->step.B, n_groups = cdiv(count, (int64_t)G);
  int rc = TG_OK;
  hipError_t e = hipSuccess;
#define TG_RUN_HIP(call)                                     \
  if (rc == TG_OK && (e = (call)) != hipSuccess) {           \
    set_hip_error(e, "tg_eval_restart_run: " #call);         \
    rc = TG_EHIP;                                            \
  }
  auto ctx_of = [&](int64_t k) { return (int)(((k / G) & 1) * G + k % G); };
  // the passes (collate-only, list form) of group q on the side stream; every count follows its pass to the host
  auto passes = [&](int64_t q) {
    for (int64_t k = q * G; k < std::min<int64_t>((q + 1) * G, count) && rc == TG_OK; ++k) {
      const int c = ctx_of(k);
      tg_step_io pio = *run->pass_io[c];
      pio.offset_dev = const_cast<int64_t*>(run->offsets + k);
      pio.advance = 0;
      rc = tg_stream_step(m, g, &pio, run->pass_ws[c], run->pass_ws_bytes[c], L->s);
      if (rc == TG_OK && run->batch_dev) hipLaunchKernelGGL(k_inc_i64, dim3(1), dim3(64), 0, L->s, run->batch_dev);
      TG_RUN_HIP(hipMemcpyAsync(run->count_host[c], pio.counts + 3, sizeof(int32_t), hipMemcpyDeviceToHost, L->s));
      TG_RUN_HIP(hipEventRecord(L->counted[c], L->s));
    }
  };
  TG_RUN_HIP(hipEventRecord(L->join, st));  // whatever the caller enqueued (the bitmap it handed over) precedes the passes
  TG_RUN_HIP(hipStreamWaitEvent(L->s, L->join, 0));
  passes(0);
  for (int64_t q = 0; q < n_groups && rc == TG_OK; ++q) {
    const int set = (int)(q & 1);
    const int64_t k0 = q * G, k1 = std::min<int64_t>(k0 + G, count);
    // the counts of this group's passes: enqueued a group ago AHEAD of the previous forward, so the host finds them there
    const int64_t* lists[TG_RESTART_MAX_LISTS];
    const float* tmins[TG_RESTART_MAX_LISTS];
    int64_t counts[TG_RESTART_MAX_LISTS], total = 0;
    for (int64_t k = k0; k < k1 && rc == TG_OK; ++k) {
      const int c = ctx_of(k);
      TG_RUN_HIP(hipEventSynchronize(L->counted[c]));
      if (rc != TG_OK) break;
      const int64_t n = *run->count_host[c];
      if (n < 0 || n > run->cap) rc = TG_EINVAL;
      if (run->n_restarted) run->n_restarted[k] = (int32_t)n;
      lists[k - k0] = run->pass_io[c]->lazy->list;
      tmins[k - k0] = run->pass_io[c]->lazy->tmin;
      counts[k - k0] = n;
      total += n;
    }
    if (rc == TG_OK && total > run->rows_cap) rc = TG_EINVAL;
    if (rc != TG_OK) break;
    // the next group's passes write the other half of the contexts: last read by the forward before this one, same stream
    if (q + 1 < n_groups) passes(q + 1);
    if (total && rc == TG_OK) {
      // row set `set` was read by the apply of group q - 2 (long done in the steady state: the side stream does not wait)
      if (q >= 2) TG_RUN_HIP(hipStreamWaitEvent(L->s, L->applied[set], 0));
      // ONE forward over the group's lists (reads the graph, the features, its parameters) beside the previous group's steps
      // (in chunks of fwd_nodes nodes - the workspace's capacity, far below the lists' bound, which only the first batches
      //  of a stream come near: a chunk is a run of whole and partial lists)
      int li = 0;
      int64_t lo = 0, done = 0;
      while (li < (int)(k1 - k0) && rc == TG_OK) {
        const int64_t* sub[TG_RESTART_MAX_LISTS];
        const float* sub_t[TG_RESTART_MAX_LISTS];
        int64_t sub_n[TG_RESTART_MAX_LISTS], room = run->fwd_nodes;
        int ns = 0;
        while (li < (int)(k1 - k0) && room > 0) {
          const int64_t take = std::min(counts[li] - lo, room);
          if (take > 0) {
            sub[ns] = lists[li] + lo;
            sub_t[ns] = tmins[li];
            sub_n[ns++] = take;
            room -= take;
            lo += take;
          }
          if (lo == counts[li]) {
            ++li;
            lo = 0;
          }
        }
        const int64_t chunk = run->fwd_nodes - room;
        if (chunk > 0)
          rc = r ? tg_restart_seq_lists_fwd(m, run->g_restart, r, ns, sub, sub_n, sub_t, run->ids[set] + done,
                                            run->h_left[set] + done * m->d, run->h_right[set] + done * m->d,
                                            run->prev_ts[set] + done, run->fwd_ws, run->fwd_ws_bytes, L->s)
                 : tg_restart_static_lists_fwd(m, run->g_restart, run->static_left, run->static_right, ns, sub, sub_n, sub_t,
                                               run->ids[set] + done, run->h_left[set] + done * m->d,
                                               run->h_right[set] + done * m->d, run->prev_ts[set] + done, L->s);
        done += chunk;
      }
      TG_RUN_HIP(hipEventRecord(L->fwd_done[set], L->s));
      // ... and the state the rows go to, on the caller's stream: the whole group at once - a node listed for batch k + 1
      // is not involved in batch k (it would have been listed there), so step k neither reads nor writes it
      TG_RUN_HIP(hipStreamWaitEvent(st, L->fwd_done[set], 0));
      if (rc == TG_OK)
        rc = tg_restart_apply(m, total, run->ids[set], run->h_left[set], run->h_right[set], run->prev_ts[set], st);
      if (rc == TG_OK && run->gtab_ws)
        rc = tg_attn_gtab_rows(m, total, run->ids[set], nullptr, run->gtab_ws, run->gtab_ws_bytes, st);
    }
    TG_RUN_HIP(hipEventRecord(L->applied[set], st));
    // collate prefetch INSIDE a group (tg_step_io.prefetch_state): between two steps of a group nothing restarts, so step k
    // may run the sampler + centres of batch k + 1 on its last launch, as in the plain resident pass; the group's last step
    // must not (the next group's apply changes the state behind a prefetched collate): it is shown a stream that ends with
    // its own batch - it consumes what step k - 1 prefetched and its rider does nothing
    int32_t pf_state = 0;
    for (int64_t k = k0; k < k1 && rc == TG_OK; ++k) {
      tg_train_io sio = *step_io;
      if (run->pos_scores) sio.pos_scores = run->pos_scores + k * B;
      if (run->neg_scores) sio.neg_scores = run->neg_scores + k * B;
      if (run->stream_len > 0) {
        sio.step.prefetch_state = &pf_state;
        sio.step.stream_len = (k + 1 < k1) ? run->stream_len : run->first_offset + (k + 1) * B;
        sio.step.l1_nids = nullptr;
        sio.step.l1_eids = nullptr;
        sio.step.l1_ts = nullptr;
      }
      rc = tg_train_step(m, g, &sio, step_ws, step_ws_bytes, st);
    }
  }
#undef TG_RUN_HIP
  // the caller's stream is ordered behind the side stream again (the bitmap the passes marked; on an error: whatever is in flight)
  if (hipEventRecord(L->join, L->s) != hipSuccess || hipStreamWaitEvent(st, L->join, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(L->s);
  }
  return rc;
}

extern "C" int tg_adam_step(const tg_adam_seg* segs_dev, int32_t n_segs, int32_t n_groups, const int32_t* enabled_dev,
                            int32_t* steps_dev, float lr, float beta1, float beta2, float eps, float grad_scale,
                            void* stream) {
  if (!segs_dev || n_segs <= 0 || n_groups <= 0 || !steps_dev) return TG_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_adam_tick, dim3((unsigned)cdiv(n_groups, 64)), dim3(64), 0, st, n_groups, enabled_dev, steps_dev);
  hipLaunchKernelGGL(k_adam, dim3(256, (unsigned)n_segs), dim3(256), 0, st, segs_dev, enabled_dev, steps_dev, lr, beta1, beta2,
                     eps, grad_scale);
  return check_launch("tg_adam_step");
}
