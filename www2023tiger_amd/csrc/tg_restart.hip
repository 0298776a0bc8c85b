// SeqRestarter forward (tiger/model/restarters.py:51-114).
//
// The reference runs a full self-attention over the last H events and then takes the
// MEAN over the H outputs.  The mean is linear, so only the column means of the
// attention matrix are needed:  mean_t(out_t) = Wo concat_h(Wv_h (sum_s abar_h[s] x_s) + bv_h) + bo
// with abar_h[s] = (1/H) sum_t A_h[t,s]  (rows of A sum to one, so bv passes through).
// Q and K still need every position (H x H scores); V and the output projection collapse
// to one row per node.  Identical maths, ~half the flops.
#include <algorithm>

#include "tg_step.h"

namespace tg {

// X[(i,t), :] = [nfeat[src] | nfeat[dst] | anony_emb[anon] | efeat[eid] | TE_r(ts_last - ts_t)],
// with the first dm-d columns of the last event zeroed (restarters.py:98-103).
__global__ void k_seq_build(tg_model m, tg_seq_restarter r, int64_t n, const int64_t* __restrict__ nids,
                            const int64_t* __restrict__ h_n, const int64_t* __restrict__ anon,
                            const int64_t* __restrict__ h_e, const float* __restrict__ h_t,
                            const int64_t* __restrict__ h_d, float4* __restrict__ X, float* __restrict__ prev_ts,
                            const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int H = r.hist_len, d4 = m.d / 4, e4 = m.d_e / 4;
  const int row4 = 4 * d4 + e4;
  const int64_t total = n * H * row4;
  const float4* nf = reinterpret_cast<const float4*>(m.nfeats);
  const float4* ef = reinterpret_cast<const float4*>(m.efeats);
  const float4* ae = reinterpret_cast<const float4*>(r.anony_emb);
  const float4* fq = reinterpret_cast<const float4*>(r.te_freq);
  const float4* ph = reinterpret_cast<const float4*>(r.te_phase);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rowi = t / row4;
    const int c = (int)(t - rowi * row4);
    const int64_t i = rowi / H;
    const int pos = (int)(rowi - i * H);
    const bool last = pos == H - 1;
    float4 v = z;
    if (c >= 3 * d4 + e4) {
      const int cc = c - 3 * d4 - e4;
      const float dt = h_t[i * H + H - 1] - h_t[rowi];
      const float4 w = fq[cc], q = ph[cc];
      v = make_float4(time_enc(dt, w.x, q.x), time_enc(dt, w.y, q.y), time_enc(dt, w.z, q.z), time_enc(dt, w.w, q.w));
    } else if (!last) {
      if (c < 2 * d4) {
        if (nf) {
          const int64_t dir = h_d[rowi];
          const int64_t self = nids[i], oth = h_n[rowi];
          // dir == 1: the query node was the destination (graph.py:239-240)
          const int64_t s_n = dir ? self : oth, d_n = dir ? oth : self;
          v = c < d4 ? nf[s_n * d4 + c] : nf[d_n * d4 + (c - d4)];
        }
      } else if (c < 3 * d4) {
        v = ae[anon[rowi] * d4 + (c - 2 * d4)];
      } else if (ef) {
        v = ef[h_e[rowi] * e4 + (c - 3 * d4)];
      }
    }
    X[t] = v;
    if (c == 0 && last) prev_ts[i] = h_t[rowi];
  }
}

// One block per (node, head): scores = q k^T / sqrt(dh) over the H x H grid, key padding
// mask, row softmax, column mean.  qk is [n*H, 2*dm] = [q | k].
template <int HMAX>
__global__ void __launch_bounds__(256) k_seq_scores(int64_t n, int H, int dm, int nh, const float* __restrict__ qk,
                                                    const int64_t* __restrict__ h_n, float* __restrict__ abar,
                                                    const int32_t* __restrict__ n_dev, DropCfg dc,
                                                    float* __restrict__ rbar) {
  if (n_dev && (int64_t)(blockIdx.x / nh) >= (int64_t)*n_dev) return;
  const uint64_t dkey = drop_key(dc);
  constexpr int CH = 32;                   // dh chunk staged per iteration
  constexpr int PPT = (HMAX * HMAX + 255) / 256;  // (t,s) pairs per thread
  __shared__ float sq[HMAX][CH + 1], sk[HMAX][CH + 1];
  __shared__ float sc[HMAX][HMAX + 1];
  __shared__ float colm[HMAX];
  const int64_t i = blockIdx.x / nh;
  const int h = blockIdx.x % nh;
  const int dh = dm / nh;
  const int tid = threadIdx.x;
  float acc[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) acc[j] = 0.f;
  const float* base = qk + (int64_t)i * H * 2 * dm + (int64_t)h * dh;
  for (int c0 = 0; c0 < dh; c0 += CH) {
    for (int f = tid; f < H * CH; f += 256) {
      const int row = f / CH, cc = f % CH;
      const bool ok = c0 + cc < dh;
      sq[row][cc] = ok ? base[(int64_t)row * 2 * dm + c0 + cc] : 0.f;
      sk[row][cc] = ok ? base[(int64_t)row * 2 * dm + dm + c0 + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int p = tid + j * 256;
      if (p < H * H) {
        const int t = p / H, s = p % H;
        float a = acc[j];
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) a += sq[t][cc] * sk[s][cc];
        acc[j] = a;
      }
    }
    __syncthreads();
  }
  const float scale = 1.0f / sqrtf((float)dh);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = tid + j * 256;
    if (p < H * H) {
      const int t = p / H, s = p % H;
      const bool masked = (s != H - 1) && (h_n[i * H + s] == 0);  // restarters.py:86-87
      sc[t][s] = masked ? -INFINITY : acc[j] * scale;
    }
  }
  __syncthreads();
  if (tid < H) {  // row softmax
    float mx = -INFINITY;
    for (int s = 0; s < H; ++s) mx = fmaxf(mx, sc[tid][s]);
    float sum = 0.f;
    for (int s = 0; s < H; ++s) {
      const float e = expf(sc[tid][s] - mx);
      sc[tid][s] = e;
      sum += e;
    }
    const float inv = 1.f / sum;
    for (int s = 0; s < H; ++s) {
      float ms = 1.f;  // attention dropout (nn.MultiheadAttention) on the normalised probabilities
      if (dc.p > 0.f)
        ms = drop_keep(dkey, DROP_SEQ_ATTN, (((uint64_t)i * nh + h) * H + tid) * H + s, dc.thresh) ? dc.scale : 0.f;
      sc[tid][s] *= inv * ms;
    }
  }
  __syncthreads();
  if (tid < H) {  // column mean
    float a = 0.f;
    for (int t = 0; t < H; ++t) a += sc[t][tid];
    a /= (float)H;
    abar[((int64_t)i * nh + h) * H + tid] = a;
    colm[tid] = a;
  }
  __syncthreads();
  if (tid == 0 && rbar) {  // sum of the column means: weight of the value bias (1 without dropout)
    float r = 0.f;
    for (int s = 0; s < H; ++s) r += colm[s];
    rbar[(int64_t)i * nh + h] = dc.p > 0.f ? r : 1.f;
  }
}

// xbar[i, h, :] = sum_s abar[i, h, s] * X[(i, s), :]
__global__ void k_seq_mix(int64_t n, int H, int row4, int nh, const float* __restrict__ abar,
                          const float4* __restrict__ X, float4* __restrict__ xbar, const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int64_t total = n * nh * row4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % row4);
    const int64_t ih = t / row4;
    const int64_t i = ih / nh;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < H; ++s) {
      const float w = abar[ih * H + s];
      const float4 x = X[((int64_t)i * H + s) * row4 + c];
      a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w;
    }
    xbar[t] = a;
  }
}

// ---------------------------------------------------------------------------------
// backward kernels (mutual loss, tiger.py:576-590)
// ---------------------------------------------------------------------------------
// dabar[i, h, s] = dxbar[i, h, :] . X[(i, s), :]   (one wavefront per (i, h, s))
__global__ void __launch_bounds__(256) k_seq_mix_bwd(int64_t n, const int32_t* __restrict__ n_dev, int H, int row4,
                                                     int nh, const float4* __restrict__ dxbar,
                                                     const float4* __restrict__ X, float* __restrict__ dabar,
                                                     const float* __restrict__ dO, const float* __restrict__ bv) {
  // dO / bv non-null (dropout): abar also weights the value bias, d rbar = dO_h . bv_h is added
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int lane = lane_id();
  const int64_t total = n * nh * H;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += (int64_t)gridDim.x * 4) {
    const int sidx = (int)(t % H);
    const int64_t ih = t / H;
    const int64_t i = ih / nh;
    const float4* a = dxbar + ih * row4;
    const float4* b = X + ((int64_t)i * H + sidx) * row4;
    float acc = 0.f;
    for (int c = lane; c < row4; c += TG_WAVE) {
      const float4 u = a[c], v = b[c];
      acc = fmaf(u.x, v.x, fmaf(u.y, v.y, fmaf(u.z, v.z, fmaf(u.w, v.w, acc))));
    }
    if (dO) {
      const int dm = row4 * 4, dh = dm / nh, h = (int)(ih % nh);
      for (int c = lane; c < dh; c += TG_WAVE) acc = fmaf(dO[i * dm + h * dh + c], bv[h * dh + c], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) dabar[t] = acc;
  }
}

// One block per (node, head): recompute the H x H attention (as k_seq_scores), then
//   dA[t, s] = dabar[s] / H;  dS[t, s] = A[t, s] (dA[t, s] - sum_s' A[t, s'] dA[t, s'])
//   dq_t = scale sum_s dS[t, s] k_s;   dk_s = scale sum_t dS[t, s] q_t
template <int HMAX>
__global__ void __launch_bounds__(256) k_seq_scores_bwd(int64_t n, const int32_t* __restrict__ n_dev, int H, int dm,
                                                        int nh, const float* __restrict__ qk,
                                                        const int64_t* __restrict__ h_n,
                                                        const float* __restrict__ dabar, float* __restrict__ dqk,
                                                        DropCfg dc) {
  if (n_dev && (int64_t)(blockIdx.x / nh) >= (int64_t)*n_dev) return;
  const uint64_t dkey = drop_key(dc);
  constexpr int CH = 32;
  constexpr int PPT = (HMAX * HMAX + 255) / 256;
  constexpr int OPT = (HMAX * CH + 255) / 256;  // (row, column) outputs per thread and chunk
  __shared__ float sq[HMAX][CH + 1], sk[HMAX][CH + 1];
  __shared__ float sc[HMAX][HMAX + 1];
  const int64_t i = blockIdx.x / nh;
  const int h = blockIdx.x % nh;
  const int dh = dm / nh;
  const int tid = threadIdx.x;
  float acc[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) acc[j] = 0.f;
  const float* base = qk + (int64_t)i * H * 2 * dm + (int64_t)h * dh;
  for (int c0 = 0; c0 < dh; c0 += CH) {
    for (int f = tid; f < H * CH; f += 256) {
      const int row = f / CH, cc = f % CH;
      const bool ok = c0 + cc < dh;
      sq[row][cc] = ok ? base[(int64_t)row * 2 * dm + c0 + cc] : 0.f;
      sk[row][cc] = ok ? base[(int64_t)row * 2 * dm + dm + c0 + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int p = tid + j * 256;
      if (p < H * H) {
        const int t = p / H, s = p % H;
        float a = acc[j];
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) a += sq[t][cc] * sk[s][cc];
        acc[j] = a;
      }
    }
    __syncthreads();
  }
  const float scale = 1.0f / sqrtf((float)dh);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int p = tid + j * 256;
    if (p < H * H) {
      const int t = p / H, s = p % H;
      const bool masked = (s != H - 1) && (h_n[i * H + s] == 0);
      sc[t][s] = masked ? -INFINITY : acc[j] * scale;
    }
  }
  __syncthreads();
  if (tid < H) {  // row softmax, then the softmax backward of that row, pre-multiplied by the score scale
    float mx = -INFINITY;
    for (int s = 0; s < H; ++s) mx = fmaxf(mx, sc[tid][s]);
    float sum = 0.f;
    for (int s = 0; s < H; ++s) {
      const float e = expf(sc[tid][s] - mx);
      sc[tid][s] = e;
      sum += e;
    }
    const float inv = 1.f / sum, invH = 1.f / (float)H;
    const float* da = dabar + ((int64_t)i * nh + h) * H;
    auto dA = [&](int s) {  // d A[t, s]: through the dropout mask of this entry
      float g = da[s] * invH;
      if (dc.p > 0.f)
        g = drop_keep(dkey, DROP_SEQ_ATTN, (((uint64_t)i * nh + h) * H + tid) * H + s, dc.thresh) ? g * dc.scale : 0.f;
      return g;
    };
    float dot = 0.f;
    for (int s = 0; s < H; ++s) {
      const float a = sc[tid][s] * inv;
      sc[tid][s] = a;
      dot = fmaf(a, dA(s), dot);
    }
    for (int s = 0; s < H; ++s) sc[tid][s] = sc[tid][s] * (dA(s) - dot) * scale;
  }
  __syncthreads();
  float* ob = dqk + (int64_t)i * H * 2 * dm + (int64_t)h * dh;
  for (int c0 = 0; c0 < dh; c0 += CH) {
    for (int f = tid; f < H * CH; f += 256) {
      const int row = f / CH, cc = f % CH;
      const bool ok = c0 + cc < dh;
      sq[row][cc] = ok ? base[(int64_t)row * 2 * dm + c0 + cc] : 0.f;
      sk[row][cc] = ok ? base[(int64_t)row * 2 * dm + dm + c0 + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < OPT; ++j) {
      const int p = tid + j * 256;
      if (p < H * CH) {
        const int row = p / CH, cc = p % CH;
        float dq = 0.f, dk = 0.f;
        for (int s = 0; s < H; ++s) {
          dq = fmaf(sc[row][s], sk[s][cc], dq);
          dk = fmaf(sc[s][row], sq[s][cc], dk);
        }
        if (c0 + cc < dh) {
          ob[(int64_t)row * 2 * dm + c0 + cc] = dq;
          ob[(int64_t)row * 2 * dm + dm + c0 + cc] = dk;
        }
      }
    }
    __syncthreads();
  }
}

// gradient of the input rows that carry parameters: the anonymised-position embedding
// (columns [2d, 3d), not for the zeroed last event) and the restarter's TimeEncode (last d
// columns).  dX = dXs (from the q/k projection) + sum_h abar_h dxbar_h (from the value mix).
__global__ void __launch_bounds__(256) k_seq_build_bwd(tg_model m, tg_seq_restarter r, int64_t n,
                                                       const int32_t* __restrict__ n_dev,
                                                       const int64_t* __restrict__ anon, const float* __restrict__ h_t,
                                                       const float* __restrict__ dXs, const float* __restrict__ abar,
                                                       const float* __restrict__ dxbar, int use_lds,
                                                       float* __restrict__ danon, float* __restrict__ dfreq,
                                                       float* __restrict__ dphase) {
  extern __shared__ float lacc[];  // [2, d] TimeEncode grads, then [(H + 1), d] embedding grads when use_lds
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int H = r.hist_len, d = m.d, dm = 4 * d + m.d_e, nh = r.n_head;
  const int nl = 2 * d + (use_lds ? (H + 1) * d : 0);
  for (int c = threadIdx.x; c < nl; c += 256) lacc[c] = 0.f;
  __syncthreads();
  float* lte = lacc;
  float* lan = lacc + 2 * d;
  const int64_t total = n * H * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % d);
    const int64_t row = t / d;
    const int64_t i = row / H;
    const int pos = (int)(row - i * H);
    float ga = dXs[row * 2 * d + c], gt = dXs[row * 2 * d + d + c];
    for (int h = 0; h < nh; ++h) {
      const float w = abar[((int64_t)i * nh + h) * H + pos];
      const float* dx = dxbar + ((int64_t)i * nh + h) * dm;
      ga = fmaf(w, dx[2 * d + c], ga);
      gt = fmaf(w, dx[3 * d + m.d_e + c], gt);
    }
    if (pos != H - 1) {
      const int64_t a = anon[row];
      if (use_lds) atomicAdd(&lan[a * d + c], ga);
      else atomicAdd(danon + a * d + c, ga);
    }
    const float dt = h_t[i * H + H - 1] - h_t[row];
    const float sn = -time_enc_sin(dt, r.te_freq[c], r.te_phase[c]) * gt;
    atomicAdd(&lte[c], sn * dt);
    atomicAdd(&lte[d + c], sn);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) {
    atomicAdd(dfreq + c, lte[c]);
    atomicAdd(dphase + c, lte[d + c]);
  }
  if (use_lds)
    for (int c = threadIdx.x; c < (H + 1) * d; c += 256) atomicAdd(danon + c, lan[c]);
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
  if (lane == 0 && nv > 0.f) {
    atomicAdd(acc + 0, se);
    atomicAdd(acc + 1, nv);
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
                              double* __restrict__ tu, int32_t* __restrict__ counts2) {
  const int64_t n = min(cap, (int64_t)*n_dev);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    counts2[0] = (int32_t)n;
    counts2[1] = (int32_t)(n * H);
  }
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
  float *x, *qk, *abar, *xbar, *o, *om, *t2, *rbar;
};

static bool carve_seq(const tg_model* m, const tg_seq_restarter* r, int64_t n, Carver& cv, SeqWs& w, bool keep_t2) {
  const size_t dm = 4 * (size_t)m->d + m->d_e, H = r->hist_len, nh = r->n_head;
  w.x = cv.take<float>(n * H * dm);
  w.qk = cv.take<float>(n * H * 2 * dm);
  w.abar = cv.take<float>(n * nh * H);
  w.xbar = cv.take<float>(n * nh * dm);
  w.o = cv.take<float>(n * dm);
  w.om = cv.take<float>(n * dm);
  w.t2 = keep_t2 ? cv.take<float>(n * (size_t)m->d) : w.om;
  w.rbar = keep_t2 ? cv.take<float>(n * nh) : nullptr;
  return cv.ok;
}

static int seq_ok(const tg_model* m, const tg_seq_restarter* r) {
  if (!m || !r || m->d <= 0 || (m->d % 4) || m->d_e <= 0 || (m->d_e % 4)) return 0;
  const int dm = 4 * m->d + m->d_e;
  // head offsets only index scalar loads / row-aligned weight blocks, so dh need not be a multiple of 4
  if (r->hist_len <= 0 || r->n_head <= 0 || dm % r->n_head) return 0;
  return 1;
}

// forward on `cap` rows of which the first *n_dev (nullable: all) are live; counts2 = {n, n*H} on device
static int seq_forward(const tg_model* m, const tg_seq_restarter* r, int64_t n, const int32_t* counts2,
                       const int64_t* nids, const int64_t* h_n, const int64_t* anon, const int64_t* h_e,
                       const float* h_t, const int64_t* h_d, float* h_left, float* h_right, float* prev_ts,
                       const SeqWs& w, hipStream_t st, const DropCfg& dc = DropCfg{}) {
  const int d = m->d, dm = 4 * m->d + m->d_e, H = r->hist_len, nh = r->n_head, dh = dm / nh;
  const int32_t* n_dev = counts2;
  const int32_t* nH_dev = counts2 ? counts2 + 1 : nullptr;
  hipLaunchKernelGGL(k_seq_build, dim3(flat_grid(n * H * (dm / 4), 256)), dim3(256), 0, st, *m, *r, n, nids, h_n, anon,
                     h_e, h_t, h_d, (float4*)w.x, prev_ts, n_dev);
  int rc;
  GemmArgs g{};
  // [q | k] = X Win[0:2dm]^T + b[0:2dm]
  g.m_cap = n * H; g.m_dev = nH_dev; g.n = 2 * dm; g.k = dm; g.a0 = ASeg{w.x, dm, dm, nullptr};
  g.w = r->in_proj_w; g.ldw = dm; g.bias = r->in_proj_b; g.c = w.qk; g.ldc = 2 * dm; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  if (H <= 40)
    hipLaunchKernelGGL((k_seq_scores<40>), dim3((unsigned)(n * nh)), dim3(256), 0, st, n, H, dm, nh, w.qk, h_n, w.abar,
                       n_dev, dc, w.rbar);
  else
    hipLaunchKernelGGL((k_seq_scores<64>), dim3((unsigned)(n * nh)), dim3(256), 0, st, n, H, dm, nh, w.qk, h_n, w.abar,
                       n_dev, dc, w.rbar);
  hipLaunchKernelGGL(k_seq_mix, dim3(flat_grid(n * nh * (dm / 4), 256)), dim3(256), 0, st, n, H, dm / 4, nh, w.abar,
                     (const float4*)w.x, (float4*)w.xbar, n_dev);
  // o[:, h] = Wv_h xbar_h + bv_h
  g = GemmArgs{};
  g.m_cap = n; g.m_dev = n_dev; g.n = dh; g.k = dm; g.a0 = ASeg{w.xbar, (int64_t)nh * dm, dm, nullptr}; g.a0_bs = dm;
  g.w = r->in_proj_w + (int64_t)2 * dm * dm; g.ldw = dm; g.w_bs = (int64_t)dh * dm;
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
  float *h_t, *sl, *sr, *prev_ts, *dsl, *dsr, *dt2, *dom, *dO, *dOm, *dxbar, *dabar, *dqk, *dXs, *acc;
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
    w.dqk = cv.take<float>(n * H * 2 * dm);
    w.dXs = cv.take<float>(n * H * 2 * d);
    if (!carve_seq(m, r, n, cv, w.seq, true)) return false;
  }
  return cv.ok;
}

size_t mutual_ws_bytes(const tg_model* m, const tg_seq_restarter* r, int64_t B) {
  const size_t n = 2 * (size_t)B, d = m->d;
  size_t b = align16(n * 8) * 4 + 32 + align16(n * d * 4) * 4 + align16(2 * n) + 16 +
             align16(tg_select_latest_workspace_bytes(n, m->n_nodes));
  if (r) {
    const size_t H = r->hist_len, dm = 4 * d + m->d_e, nh = r->n_head;
    b += align16(n * H * 8) * 4 + align16(n * H * 4) + align16(n * 4) + align16(n * d * 4) + align16(n * dm * 4) * 2 +
         align16(n * nh * dm * 4) * 2 + align16(n * nh * H * 4) + align16(n * H * 2 * dm * 4) + align16(n * H * 2 * d * 4);
    b += align16(n * H * dm * 4) + align16(n * H * 2 * dm * 4) + align16(n * nh * H * 4) + align16(n * nh * dm * 4) +
         2 * align16(n * dm * 4) + align16(n * d * 4) + align16(n * nh * 4);
  }
  return b + 256;
}

// Mutual loss and its gradients (tiger.py:574-590), after STEP 4/5 produced the targets
// h_prev_left / h_prev_right.  r != NULL: SeqRestarter; else StaticRestarter tables.
int mutual_step(const tg_model* m, const tg_tcsr* g, const tg_step_io* sio, const StepWs& sw,
                const tg_seq_restarter* r, const tg_seq_restarter* gr, const float* st_left, const float* st_right,
                float* g_left, float* g_right, float* loss_out, int32_t* flag_out, float* part, size_t part_floats,
                void* ws, size_t ws_bytes, const DropCfg& dc, hipStream_t st) {
  if (!sio->h_prev_left || !sio->h_prev_right) return TG_EINVAL;
  if (r && (!seq_ok(m, r) || r->hist_len > 64 || !gr)) return TG_EUNSUPPORTED;
  if (!r && (!st_left || !st_right || !g_left || !g_right)) return TG_EINVAL;
  const int64_t B = sio->B, n = 2 * B;
  const int d = m->d;
  Carver cv(ws, ws_bytes);
  MutualWs w{};
  if (!carve_mutual(m, r, B, cv, w)) return TG_EWORKSPACE;
  int rc;
  // ---- restart data (data_loader.py:133-165): latest occurrence of every positive node, float64 times
  hipLaunchKernelGGL(k_restart_queries, dim3(flat_grid(n, 256)), dim3(256), 0, st, B, sio->ts, sio->offset_dev, w.ts2);
  if ((rc = tg_select_latest(n, sw.nids3, w.ts2, 1, m->n_nodes, w.uniq, w.index, w.count, w.sel_ws, w.sel_bytes,
                             (void*)st)) != TG_OK)
    return rc;
  const int H = r ? r->hist_len : 1;
  hipLaunchKernelGGL(k_restart_pad, dim3(flat_grid(n, 256)), dim3(256), 0, st, n, w.count, H, w.uniq, w.index, w.ts2,
                     w.tu, w.counts2);
  hipError_t e = hipMemsetAsync(w.acc, 0, 4 * sizeof(float), st);
  if (e != hipSuccess) {
    set_hip_error(e, "mutual_step memset");
    return TG_EHIP;
  }
  auto F = [](const float* p) { return const_cast<float*>(p); };
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
  hipLaunchKernelGGL(k_mutual_a, dim3(std::min<unsigned>(flat_grid(2 * n, 4), 512)), dim3(256), 0, st, n, w.counts2, d,
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
  const int32_t* nH_dev = w.counts2 + 1;
  const SeqWs& q = w.seq;
  TnArgs tn{};
  GemmArgs ga{};
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
  if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
  ga = GemmArgs{};
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = d; ga.k = d; ga.a0 = ASeg{w.dsr, d, d, nullptr};
  ga.w = r->fc2.w; ga.ldw = d; ga.w_kmajor = 1; ga.c = w.dt2; ga.ldc = d; ga.nbatch = 1;
  ga.alpha = dc.p > 0.f ? dc.scale : 1.f;  // q.t2 is the dropped activation: > 0 iff kept and positive
  ga.relu_mask = q.t2; ga.ld_mask = d;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  tn = tn_base(n, n_dev);
  tn.n = d; tn.k = d; tn.y = w.dt2; tn.ldy = d; tn.x0 = ASeg{w.sl, d, d, nullptr};
  tn.out = F(gr->fc1.w); tn.ldo = dm; tn.bias_out = F(gr->fc1.b);
  if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
  ga = GemmArgs{};  // d h_left += dt2 fc1[:, :d]
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = d; ga.k = d; ga.a0 = ASeg{w.dt2, d, d, nullptr};
  ga.w = r->fc1.w; ga.ldw = dm; ga.w_kmajor = 1; ga.c = w.dsl; ga.ldc = d; ga.alpha = 1.f; ga.nbatch = 1; ga.accumulate = 1;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  // out_fn
  tn = tn_base(n, n_dev);
  tn.n = d; tn.k = dm; tn.y = w.dsl; tn.ldy = d; tn.x0 = ASeg{q.om, dm, dm, nullptr};
  tn.out = F(gr->out_fn.w); tn.ldo = dm; tn.bias_out = F(gr->out_fn.b);
  if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
  ga = GemmArgs{};
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = dm; ga.k = d; ga.a0 = ASeg{w.dsl, d, d, nullptr};
  ga.w = r->out_fn.w; ga.ldw = dm; ga.w_kmajor = 1; ga.c = w.dom; ga.ldc = dm; ga.alpha = 1.f; ga.nbatch = 1;
  ga.relu_mask = q.om; ga.ld_mask = dm;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  // out_proj
  tn = tn_base(n, n_dev);
  tn.n = dm; tn.k = dm; tn.y = w.dom; tn.ldy = dm; tn.x0 = ASeg{q.o, dm, dm, nullptr};
  tn.out = F(gr->out_proj.w); tn.ldo = dm; tn.bias_out = F(gr->out_proj.b);
  if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
  ga = GemmArgs{};
  ga.m_cap = n; ga.m_dev = n_dev; ga.n = dm; ga.k = dm; ga.a0 = ASeg{w.dom, dm, dm, nullptr};
  ga.w = r->out_proj.w; ga.ldw = dm; ga.w_kmajor = 1; ga.c = w.dO; ga.ldc = dm; ga.alpha = 1.f; ga.nbatch = 1;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  // value projection per head.  dh = dm / nh need not be a multiple of 4 (d = 172: dh = 430), so the
  // head slices of dO cannot be addressed as aligned sub-matrices; instead each head uses a copy of
  // dO with the other heads' columns zeroed and full-width (K = dm) products.
  hipLaunchKernelGGL(k_head_mask, dim3(flat_grid(n * dm, 256)), dim3(256), 0, st, n, n_dev, dm, nh, w.dO, w.dOm);
  for (int h = 0; h < nh; ++h) {
    const float* dOh = w.dOm + (int64_t)h * n * dm;
    tn = tn_base(n, n_dev);
    tn.n = dm; tn.k = dm; tn.y = dOh; tn.ldy = dm; tn.x0 = ASeg{q.xbar + (int64_t)h * dm, (int64_t)nh * dm, dm, nullptr};
    tn.out = F(gr->in_proj_w) + (int64_t)2 * dm * dm; tn.ldo = dm; tn.bias_out = F(gr->in_proj_b) + 2 * dm;
    if (dc.p > 0.f) { tn.bias_rs = q.rbar; tn.ld_brs = nh; tn.brs_col = h; }
    if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
    ga = GemmArgs{};
    ga.m_cap = n; ga.m_dev = n_dev; ga.n = dm; ga.k = dm; ga.a0 = ASeg{dOh, dm, dm, nullptr};
    ga.w = r->in_proj_w + (int64_t)2 * dm * dm; ga.ldw = dm; ga.w_kmajor = 1;
    ga.c = w.dxbar + (int64_t)h * dm; ga.ldc = (int64_t)nh * dm; ga.alpha = 1.f; ga.nbatch = 1;
    if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  }
  // value mix and attention scores
  hipLaunchKernelGGL(k_seq_mix_bwd, dim3(flat_grid(n * nh * H, 4)), dim3(256), 0, st, n, n_dev, H, dm / 4, nh,
                     (const float4*)w.dxbar, (const float4*)q.x, w.dabar, dc.p > 0.f ? w.dO : (const float*)nullptr,
                     r->in_proj_b + 2 * dm);
  if (H <= 40)
    hipLaunchKernelGGL((k_seq_scores_bwd<40>), dim3((unsigned)(n * nh)), dim3(256), 0, st, n, n_dev, H, dm, nh, q.qk,
                       w.h_n, w.dabar, w.dqk, dc);
  else
    hipLaunchKernelGGL((k_seq_scores_bwd<64>), dim3((unsigned)(n * nh)), dim3(256), 0, st, n, n_dev, H, dm, nh, q.qk,
                       w.h_n, w.dabar, w.dqk, dc);
  // q/k projection
  tn = tn_base(n * H, nH_dev);
  tn.n = 2 * dm; tn.k = dm; tn.y = w.dqk; tn.ldy = 2 * dm; tn.x0 = ASeg{q.x, dm, dm, nullptr};
  tn.out = F(gr->in_proj_w); tn.ldo = dm; tn.bias_out = F(gr->in_proj_b);
  if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
  // input-row gradients, only for the two column blocks that carry parameters
  ga = GemmArgs{};
  ga.m_cap = n * H; ga.m_dev = nH_dev; ga.n = d; ga.k = 2 * dm; ga.a0 = ASeg{w.dqk, 2 * dm, 2 * dm, nullptr};
  ga.w = r->in_proj_w + 2 * d; ga.ldw = dm; ga.w_kmajor = 1; ga.c = w.dXs; ga.ldc = 2 * d; ga.alpha = 1.f; ga.nbatch = 1;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  ga.w = r->in_proj_w + 3 * d + m->d_e; ga.c = w.dXs + d;
  if ((rc = gemm_launch(ga, st)) != TG_OK) return rc;
  const size_t lfull = (size_t)(2 * d + (H + 1) * d) * sizeof(float);
  const int use_lds = lfull <= 60 * 1024;
  hipLaunchKernelGGL(k_seq_build_bwd, dim3(std::min<unsigned>(flat_grid(n * H * d, 256), 512)), dim3(256),
                     use_lds ? lfull : (size_t)2 * d * sizeof(float), st, *m, *r, n, n_dev, w.anon, w.h_t, w.dXs, q.abar,
                     w.dxbar, use_lds, F(gr->anony_emb), F(gr->te_freq), F(gr->te_phase));
  return check_launch("mutual_step(seq)");
}

}  // namespace tg

using namespace tg;

extern "C" size_t tg_restart_seq_workspace_bytes(const tg_model* m, const tg_seq_restarter* r, int64_t n) {
  if (!seq_ok(m, r) || n < 0) return 0;
  const size_t dm = 4 * (size_t)m->d + m->d_e, H = r->hist_len, nh = r->n_head;
  return align16(n * H * dm * 4) + align16(n * H * 2 * dm * 4) + align16(n * nh * H * 4) + align16(n * nh * dm * 4) +
         2 * align16(n * dm * 4) + align16(n * (size_t)m->d * 4) + align16(n * nh * 4) + 64;
}

extern "C" int tg_restart_seq_fwd(const tg_model* m, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                                  const int64_t* h_n, const int64_t* anon, const int64_t* h_e, const float* h_t,
                                  const int64_t* h_d, float* h_left, float* h_right, float* prev_ts, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!seq_ok(m, r) || n < 0) return TG_EINVAL;
  if (r->hist_len > 64) return TG_EUNSUPPORTED;
  if (n == 0) return TG_OK;
  if (!nids || !h_n || !anon || !h_e || !h_t || !h_d || !h_left || !h_right || !prev_ts) return TG_EINVAL;
  Carver cv(ws, ws_bytes);
  SeqWs w{};
  if (!ws || !carve_seq(m, r, n, cv, w, false)) return TG_EWORKSPACE;
  return seq_forward(m, r, n, nullptr, nids, h_n, anon, h_e, h_t, h_d, h_left, h_right, prev_ts, w, as_stream(stream));
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
  if (r->hist_len > 64) return TG_EUNSUPPORTED;
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
