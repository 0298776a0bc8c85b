// tg_attn_fuse: products of attention PARAMETERS for inference with fixed weights (see
// include/tiger_hip.h).  Reference maths: temporal_agg_modules.py:204-235 (MultiheadAttention
// + MergeLayer); re-associated so that the forward pass multiplies activations three times
// instead of six.  Runs once per parameter version, so nothing here is tuned.
#include "tg_step.h"
#include "tg_tile.h"

namespace tg {

// qconst[n] = bq[n] + sum_j Wq[n, d + j] cos(phi_j)                    (TE(0) = cos(phi))
// c0[n]     = bo[n] + sum_k Wo[n, k] bv[k]
__global__ void k_fuse_vec1(int d, const float* __restrict__ wq, const float* __restrict__ b_in,
                            const float* __restrict__ freq, const float* __restrict__ phase,
                            const float* __restrict__ wo, const float* __restrict__ bo, float* __restrict__ qconst,
                            float* __restrict__ c0) {
  const int E = 2 * d, lane = lane_id();
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * E) return;
  const int n = row % E;
  float acc = 0.f;
  if (row < E) {
    for (int j = lane; j < d; j += TG_WAVE) acc += wq[(int64_t)n * E + d + j] * time_enc(0.f, freq[j], phase[j]);
    acc = wave_sum(acc);
    if (lane == 0) qconst[n] = acc + b_in[n];
  } else {
    for (int k = lane; k < E; k += TG_WAVE) acc += wo[(int64_t)n * E + k] * b_in[2 * E + k];
    acc = wave_sum(acc);
    if (lane == 0) c0[n] = acc + bo[n];
  }
}
// gconst[h kvw + c] = alpha sum_k Wk_h[k, c] qconst[h dh + k];  c1[n] = sum_e W1[n, e] c0[e];  b1 copy;
// W1f[n, nk + j] = W1[n, E + j]
__global__ void k_fuse_vec2(int d, int kvw, int nh, float alpha, const float* __restrict__ wk,
                            const float* __restrict__ qconst, const float* __restrict__ w1, const float* __restrict__ b1,
                            const float* __restrict__ c0, float* __restrict__ gconst, float* __restrict__ w1f,
                            float* __restrict__ b1o, float* __restrict__ c1) {
  const int E = 2 * d, dh = E / nh, nk = nh * kvw;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nk) {
    const int h = t / kvw, c = t % kvw;
    float acc = 0.f;
    for (int k = 0; k < dh; ++k) acc += wk[((int64_t)h * dh + k) * kvw + c] * qconst[h * dh + k];
    gconst[t] = alpha * acc;
  }
  if (t < d) {
    float acc = 0.f;
    for (int e = 0; e < E; ++e) acc += w1[(int64_t)t * (E + d) + e] * c0[e];
    c1[t] = acc;
    b1o[t] = b1[t];
  }
  for (int e = t; e < d * d; e += gridDim.x * blockDim.x) {
    const int n = e / d, j = e % d;
    w1f[(int64_t)n * (nk + d) + nk + j] = w1[(int64_t)n * (E + d) + E + j];
  }
}

// Without an edge table the edge segment of every key row is zeros (feature_getter.py:95-99), so the columns of Wqk /
// gconst and of W1f that meet it multiply zeros: the compact form drops them (kvw -> 2d per head) and the attention
// block runs a third fewer flops.  full: [Wqk [nk, d] | gconst [nk] | W1f [d, nk + d] | b1 | c1], nk = nh kvw;
// out: the same with nkc = nh (kvw - de).
__global__ void k_fuse_compact(int d, int de, int nh, const float* __restrict__ full, float* __restrict__ out) {
  const int kvw = 2 * d + de, kc = 2 * d, nk = nh * kvw, nkc = nh * kc;
  auto src_col = [&](int cc) {  // compact key column -> full key column
    const int h = cc / kc, c = cc % kc;
    return h * kvw + (c < d ? c : c + de);
  };
  const float* wqk = full;
  const float* gconst = wqk + (size_t)nk * d;
  const float* w1f = gconst + nk;
  const float* tailv = w1f + (size_t)d * (nk + d);  // b1 | c1
  float* o_wqk = out;
  float* o_gconst = o_wqk + (size_t)nkc * d;
  float* o_w1f = o_gconst + nkc;
  float* o_tail = o_w1f + (size_t)d * (nkc + d);
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = t0; t < (int64_t)nkc * d; t += nth) o_wqk[t] = wqk[(size_t)src_col((int)(t / d)) * d + t % d];
  for (int64_t t = t0; t < nkc; t += nth) o_gconst[t] = gconst[src_col((int)t)];
  for (int64_t t = t0; t < (int64_t)d * (nkc + d); t += nth) {
    const int n = (int)(t / (nkc + d)), c = (int)(t % (nkc + d));
    o_w1f[t] = w1f[(size_t)n * (nk + d) + (c < nkc ? src_col(c) : nk + (c - nkc))];
  }
  for (int64_t t = t0; t < 2 * d; t += nth) o_tail[t] = tailv[t];
}

size_t attn_fused_floats_of(size_t d, size_t nk) { return nk * d + nk + d * (nk + d) + 2 * d; }

// ---- the blob's tail for the split updater (tg_dense.h: GruTail): [W2 ; W_hh W2] [4d, d], then [b2 ; W_hh b2 + b_hh] [4d]
// (GRUCell weight_hh / bias_hh, update_modules.py:33; the merger's fc2, basic_modules.py:16-19).  Present whenever the model
// has a GRU updater; tg_stream_step decides whether the step can use it.
static size_t gru_tail_floats(const tg_model* m) {
  return (m->upd_fn == TG_UPD_GRU && m->gru_w_hh && m->gru_b_hh && m->attn_fc2.w && m->attn_fc2.b) ? (size_t)4 * m->d * (m->d + 1) : 0;
}
static size_t gru_tail_offset(const tg_model* m) {
  const size_t d = m->d, nk = (size_t)m->n_head * (2 * d + (m->efeats ? m->d_e : 0));
  return attn_fused_floats_of(d, nk) + (tile_waves_for_shape(m) ? tile_dims(m).floats : 0);
}
const float* gru_tail_weights(const tg_model* m) {
  return (m->attn_fused && gru_tail_floats(m)) ? m->attn_fused + gru_tail_offset(m) : nullptr;
}
__global__ void k_tail_bias(int d, const float* __restrict__ w_hh, const float* __restrict__ b_hh, const float* __restrict__ b2,
                            float* __restrict__ out) {
  const int lane = lane_id();
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);  // 0 .. 4d - 1
  if (row >= 4 * d) return;
  if (row < d) {
    if (lane == 0) out[row] = b2[row];
    return;
  }
  const int r = row - d;
  float acc = 0.f;
  for (int k = lane; k < d; k += TG_WAVE) acc += w_hh[(int64_t)r * d + k] * b2[k];
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc + b_hh[r];
}

// fragment-major copy of K columns [k_src0, k_src0 + K) of W[N][ldw] into k-chunks [kc_off, kc_off + ceil(K / 16)) of a
// packed weight with KC chunks per column tile (tg_tile.h); zeros where n >= N or k >= K
__global__ void k_pack_frag(const float* __restrict__ w, int N, int K, int64_t ldw, int k_src0, float* __restrict__ out,
                            int NT, int KC, int kc_off) {
  const int kcs = (K + 15) / 16;
  const int64_t total = (int64_t)NT * kcs * 256;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(e & 3), l = (int)((e >> 2) & 63);
    const int64_t blk = e >> 8;
    const int kc = (int)(blk % kcs), nt = (int)(blk / kcs);
    const int n = nt * 16 + (l & 15), k = kc * 16 + 4 * (l >> 4) + j;
    out[((int64_t)nt * KC + kc_off + kc) * 256 + l * 4 + j] = (n < N && k < K) ? w[(int64_t)n * ldw + k_src0 + k] : 0.f;
  }
}
__global__ void k_pad_vec(const float* __restrict__ v, int n, float* __restrict__ out, int np) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) out[i] = i < n ? v[i] : 0.f;
}

}  // namespace tg

using namespace tg;

extern "C" size_t tg_attn_fused_floats(const tg_model* m) {
  if (!attn_dims_ok(m)) return 0;
  const size_t d = m->d, nk = (size_t)m->n_head * (2 * d + (m->efeats ? m->d_e : 0));  // compact without an edge table
  // + the tile form of the same weights and of fc2 (tg_tile.h) where the one-launch attention applies
  return attn_fused_floats_of(d, nk) + (tile_waves_for_shape(m) ? tile_dims(m).floats : 0) + gru_tail_floats(m);
}

extern "C" int tg_attn_tile_applies(const tg_model* m) { return (m && attn_dims_ok(m)) ? attn_tile_applies(m) : 0; }

extern "C" size_t tg_attn_fuse_workspace_bytes(const tg_model* m) {
  if (!attn_dims_ok(m)) return 0;
  const size_t d = m->d, E = 2 * d, nk = (size_t)m->n_head * (2 * d + m->d_e);
  // qconst [E], c0 [E], Wov [E, nk], partials of the weight-gradient-style product, the full form before compaction
  return align16(E * 4) * 2 + align16(E * nk * 4) + align16((size_t)16 * nk * (d + 1) * 4) +
         align16(attn_fused_floats_of(d, nk) * 4) + 256;
}

extern "C" int tg_attn_fuse(const tg_model* m, float* fused, void* ws, size_t ws_bytes, void* stream) {
  if (!attn_dims_ok(m) || !fused) return TG_EINVAL;
  hipStream_t st = as_stream(stream);
  const int d = m->d, E = 2 * d, kvw = 2 * d + m->d_e, nh = m->n_head, dh = E / nh, nk = nh * kvw;
  const float alpha = 1.0f / sqrtf((float)dh);
  Carver cv(ws, ws_bytes);
  float* qconst = cv.take<float>(E);
  float* c0 = cv.take<float>(E);
  float* wov = cv.take<float>((size_t)E * nk);
  const size_t part_floats = (size_t)16 * nk * (d + 1);
  float* part = cv.take<float>(part_floats);
  const bool compact = m->efeats == nullptr;  // built in full into scratch, then the edge columns are dropped
  float* full = compact ? cv.take<float>(attn_fused_floats_of(d, nk)) : fused;
  if (!cv.ok) return TG_EWORKSPACE;
  float* wqk = full;
  float* gconst = wqk + (size_t)nk * d;
  float* w1f = gconst + nk;
  float* b1 = w1f + (size_t)d * (nk + d);
  float* c1 = b1 + d;
  hipLaunchKernelGGL(k_fuse_vec1, dim3((unsigned)cdiv(2 * E, 4)), dim3(256), 0, st, d, m->attn_wq, m->attn_b_in, m->te_freq,
                     m->te_phase, m->attn_out.w, m->attn_out.b, qconst, c0);
  int rc;
  // Wqk_h[c, j] = alpha sum_k Wk_h[k, c] Wq[h dh + k, j]: both operands are read along k, i.e. the
  // weight-gradient product shape
  TnArgs tn{};
  tn.m_cap = dh; tn.n = kvw; tn.k = d; tn.y = m->attn_wk; tn.ldy = kvw; tn.y_bs = (int64_t)dh * kvw;
  tn.x0 = ASeg{m->attn_wq, E, d, nullptr}; tn.x0_bs = (int64_t)dh * E;
  tn.out = wqk; tn.ldo = d; tn.out_bs = (int64_t)kvw * d; tn.alpha = alpha; tn.accumulate = 0; tn.nbatch = nh;
  tn.part = part; tn.part_floats = part_floats;
  if ((rc = gemm_tn_launch(tn, st)) != TG_OK) return rc;
  // Wov[n, h kvw + c] = sum_k Wo[n, h dh + k] Wv_h[k, c]
  GemmArgs g{};
  g.m_cap = E; g.n = kvw; g.k = dh; g.a0 = ASeg{m->attn_out.w, E, dh, nullptr}; g.a0_bs = dh;
  g.w = m->attn_wv; g.ldw = kvw; g.w_kmajor = 1; g.w_bs = (int64_t)dh * kvw;
  g.c = wov; g.ldc = nk; g.c_bs = kvw; g.alpha = 1.f; g.nbatch = nh;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  // W1f[:, :nk] = W1[:, :E] Wov
  g = GemmArgs{};
  g.m_cap = d; g.n = nk; g.k = E; g.a0 = ASeg{m->attn_fc1.w, E + d, E, nullptr};
  g.w = wov; g.ldw = nk; g.w_kmajor = 1; g.c = w1f; g.ldc = nk + d; g.alpha = 1.f; g.nbatch = 1;
  if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
  hipLaunchKernelGGL(k_fuse_vec2, dim3((unsigned)cdiv(std::max(nk, d * d / 8), 256)), dim3(256), 0, st, d, kvw, nh, alpha,
                     m->attn_wk, qconst, m->attn_fc1.w, m->attn_fc1.b, c0, gconst, w1f, b1, c1);
  if (compact)
    hipLaunchKernelGGL(k_fuse_compact, dim3(256), dim3(256), 0, st, d, (int)m->d_e, nh, (const float*)full, fused);
  if (tile_waves_for_shape(m)) {  // the same weights (and fc2) fragment-major for k_attn_tile
    const TileDims t = tile_dims(m);
    const float* f_wqk = fused;
    const float* f_gconst = f_wqk + (size_t)t.nk * d;
    const float* f_w1f = f_gconst + t.nk;
    const float* f_b1 = f_w1f + (size_t)d * (t.nk + d);
    const float* f_c1 = f_b1 + d;
    float* tile = fused + attn_fused_floats_of(d, t.nk);
    auto pack = [&](const float* w, int N, int K, int64_t ldw, int k0, float* out, int NT, int KC, int kc_off) {
      hipLaunchKernelGGL(k_pack_frag, dim3(256), dim3(256), 0, st, w, N, K, ldw, k0, out, NT, KC, kc_off);
    };
    auto pad = [&](const float* v, int n, float* out, int np) {
      hipLaunchKernelGGL(k_pad_vec, dim3(4), dim3(256), 0, st, v, n, out, np);
    };
    pack(f_wqk, t.nk, d, d, 0, tile + t.o_wqk, t.NTg, t.KCd, 0);
    pad(f_gconst, t.nk, tile + t.o_gconst, t.nk_p);
    pack(f_w1f, d, t.nk, t.nk + d, 0, tile + t.o_w1f, t.NTd, t.KCnk + t.KCd, 0);      // the S part of [S | c]
    pack(f_w1f, d, d, t.nk + d, t.nk, tile + t.o_w1f, t.NTd, t.KCnk + t.KCd, t.KCnk);  // the c part, from a fresh chunk
    pad(f_b1, d, tile + t.o_b1, t.d_p);
    pad(f_c1, d, tile + t.o_c1, t.d_p);
    pack(m->attn_fc2.w, d, d, d, 0, tile + t.o_w2, t.NTd, t.KCd, 0);
    pad(m->attn_fc2.b, d, tile + t.o_b2, t.d_p);
    if ((rc = attn_tile_prepare()) != TG_OK) return rc;
  }
  if (gru_tail_floats(m)) {
    float* wt = fused + gru_tail_offset(m);
    float* bt = wt + (size_t)4 * d * d;
    hipError_t e = hipMemcpyAsync(wt, m->attn_fc2.w, (size_t)d * d * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
      set_hip_error(e, "tg_attn_fuse (tail)");
      return TG_EHIP;
    }
    g = GemmArgs{};  // (W_hh W2)[m, n] = sum_k W_hh[m, k] W2[k, n]: the k-major view of W2
    g.m_cap = 3 * d; g.n = d; g.k = d; g.a0 = ASeg{m->gru_w_hh, d, d, nullptr};
    g.w = m->attn_fc2.w; g.ldw = d; g.w_kmajor = 1; g.c = wt + (size_t)d * d; g.ldc = d; g.alpha = 1.f; g.nbatch = 1;
    if ((rc = gemm_launch(g, st)) != TG_OK) return rc;
    hipLaunchKernelGGL(k_tail_bias, dim3((unsigned)cdiv(4 * d, 4)), dim3(256), 0, st, d, m->gru_w_hh, m->gru_b_hh, m->attn_fc2.b, bt);
  }
  return check_launch("tg_attn_fuse");
}
