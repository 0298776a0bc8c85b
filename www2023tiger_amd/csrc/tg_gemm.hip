// float32 MFMA GEMM with gathered A rows, and the fused GRU cell (a14, a15).
// Reference: torch.nn.Linear / nn.GRUCell as used at tiger/model/update_modules.py:30-47,
// message_modules.py:29-55, basic_modules.py:16-19.
//
// Tiling (CDNA4): 256-thread blocks = 4 wavefronts, each wavefront owns 32x32 output
// tiles computed with v_mfma_f32_32x32x2_f32 (lane l feeds A[l&31][l>>5], B[l>>5][l&31]).
// Operand tiles are staged global -> registers -> LDS as [row][k] with a 33-float row
// stride, which makes both the scalar ds_write of a float4 and the per-lane ds_read_b32
// of the MFMA operands bank-conflict free.  Two LDS buffers, one barrier per K step;
// the next tile's global loads are issued before the MFMAs of the current one.
// blockIdx -> tile mapping keeps all N-tiles (and batches) of one M-tile on the same
// XCD (blocks b and b+8 share an L2), so gathered A rows are fetched from HBM once.
#include "tg_dense.h"

namespace tg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDK = BK + 1;

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void sts4(float* row, int k, float4 v) {
  row[k] = v.x;
  row[k + 1] = v.y;
  row[k + 2] = v.z;
  row[k + 3] = v.w;
}

template <int WM, int WN>
__global__ void __launch_bounds__(256) k_gemm(GemmArgs g) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
  __shared__ float As[2][BM][LDK];
  __shared__ float Bs[2][BN][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = (g.n + BN - 1) / BN;
  const int per = NT * g.nbatch;
  const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
  const int64_t mt = (int64_t)(s / per) * 8 + xcd;
  const int rem = s % per;
  const int nt = rem % NT, bz = rem / NT;
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  const int64_t m0 = mt * BM;
  if (m0 >= M) return;
  const int n0 = nt * BN;
  const float* a0p = g.a0.p + (int64_t)bz * g.a0_bs;
  const float* wp = g.w + (int64_t)bz * g.w_bs;
  const int K = g.k, N = g.n, kw0 = g.a0.w;

  const int ar = tid >> 3, ac4 = (tid & 7) * 4;  // A (and row-major W) tile coordinates
  int64_t row0[WM], row1[WM];
  bool rok[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int64_t m = m0 + ar + i * 32;
    rok[i] = m < M;
    row0[i] = (rok[i] && g.a0.idx) ? g.a0.idx[m] : m;
    row1[i] = (rok[i] && g.a1.p && g.a1.idx) ? g.a1.idx[m] : m;
  }
  float4 ra[WM], rb[WN];
  auto load = [&](int kt) {
    const int k = kt * BK + ac4;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      float4 v = zero4();
      if (rok[i] && k < K) v = (k < kw0) ? ldg4(a0p + row0[i] * g.a0.ld + k) : ldg4(g.a1.p + row1[i] * g.a1.ld + (k - kw0));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      float4 v = zero4();
      if (!g.w_kmajor) {
        const int n = n0 + ar + i * 32;
        if (n < N && k < K) v = ldg4(wp + (int64_t)n * g.ldw + k);
      } else {
        const int f = tid + i * 256;
        const int kk = kt * BK + f / (BN / 4), n = n0 + (f % (BN / 4)) * 4;
        if (kk < K && n < N) v = ldg4(wp + (int64_t)kk * g.ldw + n);
      }
      rb[i] = v;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WM; ++i) sts4(As[buf][ar + i * 32], ac4, ra[i]);
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      if (!g.w_kmajor) {
        sts4(Bs[buf][ar + i * 32], ac4, rb[i]);
      } else {
        const int f = tid + i * 256;
        const int kk = f / (BN / 4), nn = (f % (BN / 4)) * 4;
        Bs[buf][nn][kk] = rb[i].x;
        Bs[buf][nn + 1][kk] = rb[i].y;
        Bs[buf][nn + 2][kk] = rb[i].z;
        Bs[buf][nn + 3][kk] = rb[i].w;
      }
    }
  };
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 31, fk = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nkt = (K + BK - 1) / BK;
  load(0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    store(buf);
    __syncthreads();
    if (kt + 1 < nkt) load(kt + 1);
    const float* ap = &As[buf][wm * 32 + fr][fk];
    const float* bp = &Bs[buf][wn * 32 + fr][fk];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc, 0, 0, 0);
  }
  const int n = n0 + wn * 32 + fr;
  if (n >= N) return;
  const float bias = g.bias ? g.bias[(int64_t)bz * g.bias_bs + n] : 0.f;
  float* cp = g.c + (int64_t)bz * g.c_bs;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
    if (m >= M) continue;
    float v = g.alpha * (acc[r] + bias);
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.row_valid && !g.row_valid[m]) v = 0.f;
    const int64_t cr = g.c_rows ? (int64_t)g.c_rows[m] : m;
    cp[cr * g.ldc + n] = v;
  }
}

int gemm_launch(const GemmArgs& g, hipStream_t st) {
  if (g.m_cap <= 0) return TG_OK;
  if (g.n <= 0 || g.k <= 0 || (g.k % 4) || (g.a0.w % 4) || (g.ldw % 4) || g.nbatch <= 0) return TG_EINVAL;
  if (g.w_kmajor && (g.n % 4)) return TG_EINVAL;
  if (g.a0.w + (g.a1.p ? g.a1.w : 0) != g.k) return TG_EINVAL;
  constexpr int BM = 64, BN = 64;
  const int64_t MT = cdiv(g.m_cap, BM);
  const int NT = (int)cdiv(g.n, BN);
  const int64_t grid = 8 * cdiv(MT, 8) * NT * g.nbatch;
  hipLaunchKernelGGL((k_gemm<2, 2>), dim3((unsigned)grid), dim3(256), 0, st, g);
  return check_launch("gemm");
}

// ---------------------------------------------------------------------------------
// GRU cell, gates fused into the GEMM epilogue.  A block owns 128 rows x 32 hidden
// columns and accumulates four planes per column: r and z over K = [x | h], i_n over x
// only, h_n over h only (no wasted MFMAs on the zero blocks of a packed [4d, 5d] weight).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gru(GruArgs g) {
  constexpr int BM = 128;
  __shared__ float As[2][BM][LDK];
  __shared__ float Bs[2][3][32][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = g.d, xw = g.xw;
  const int NT = (d + 31) / 32;
  const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
  const int64_t mt = (int64_t)(s / NT) * 8 + xcd;
  const int nt = s % NT;
  int64_t M = g.cap;
  if (g.n_dev) M = min(M, (int64_t)*g.n_dev);
  const int64_t m0 = mt * BM;
  if (m0 >= M) return;
  const int j0 = nt * 32;
  const int ar = tid >> 3, ac4 = (tid & 7) * 4;
  int64_t rx[4], rh[4];
  bool rok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + ar + i * 32;
    rok[i] = m < M;
    rx[i] = (rok[i] && g.x.idx) ? g.x.idx[m] : m;
    rh[i] = (rok[i] && g.h.idx) ? g.h.idx[m] : m;
  }
  const int nkx = (xw + BK - 1) / BK, nkh = (d + BK - 1) / BK;
  const int nkt = nkx + nkh;
  float4 ra[4], rb[3];
  auto load = [&](int t) {
    const bool hp = t >= nkx;
    const int k = (hp ? t - nkx : t) * BK + ac4;
    const int width = hp ? d : xw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = zero4();
      if (rok[i] && k < width) v = hp ? ldg4(g.h.p + rh[i] * g.h.ld + k) : ldg4(g.x.p + rx[i] * g.x.ld + k);
      ra[i] = v;
    }
    const int j = j0 + ar;
    const float* wbase = hp ? g.w_hh : g.w_ih;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float4 v = zero4();
      if (j < d && k < width) v = ldg4(wbase + ((int64_t)i * d + j) * width + k);
      rb[i] = v;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sts4(As[buf][ar + i * 32], ac4, ra[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) sts4(Bs[buf][i][ar], ac4, rb[i]);
  };
  const int fr = lane & 31, fk = lane >> 5;
  f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc_r[i] = acc_z[i] = acc_in[i] = acc_hn[i] = 0.f;
  load(0);
  for (int t = 0; t < nkt; ++t) {
    const int buf = t & 1;
    store(buf);
    __syncthreads();
    if (t + 1 < nkt) load(t + 1);
    const float* ap = &As[buf][wave * 32 + fr][fk];
    const float* b0 = &Bs[buf][0][fr][fk];
    const float* b1 = &Bs[buf][1][fr][fk];
    const float* b2 = &Bs[buf][2][fr][fk];
    if (t < nkx) {
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        const float a = ap[2 * ks];
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[2 * ks], acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[2 * ks], acc_z, 0, 0, 0);
        acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2[2 * ks], acc_in, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        const float a = ap[2 * ks];
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[2 * ks], acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[2 * ks], acc_z, 0, 0, 0);
        acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2[2 * ks], acc_hn, 0, 0, 0);
      }
    }
  }
  const int j = j0 + fr;
  if (j >= d) return;
  const float br = g.b_ih[j] + g.b_hh[j];
  const float bz = g.b_ih[d + j] + g.b_hh[d + j];
  const float bin = g.b_ih[2 * d + j], bhn = g.b_hh[2 * d + j];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
    if (m >= M) continue;
    const int64_t hr = g.h.idx ? g.h.idx[m] : m;
    const float hold = g.h.p[hr * g.h.ld + j];
    const float rg = sigmoidf_(acc_r[r] + br);
    const float zg = sigmoidf_(acc_z[r] + bz);
    const float ng = tanhf(acc_in[r] + bin + rg * (acc_hn[r] + bhn));
    const int64_t orow = g.out_rows ? (int64_t)g.out_rows[m] : m;
    g.out[orow * g.ldo + j] = (1.f - zg) * ng + zg * hold;
  }
}

int gru_launch(const GruArgs& g, hipStream_t st) {
  if (g.cap <= 0) return TG_OK;
  if (g.d <= 0 || (g.d % 4) || g.xw <= 0 || (g.xw % 4)) return TG_EINVAL;
  const int64_t MT = cdiv(g.cap, 128);
  const int NT = (g.d + 31) / 32;
  const int64_t grid = 8 * cdiv(MT, 8) * NT;
  hipLaunchKernelGGL(k_gru, dim3((unsigned)grid), dim3(256), 0, st, g);
  return check_launch("gru");
}

}  // namespace tg
