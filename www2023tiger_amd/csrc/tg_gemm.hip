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
#include <cstdlib>
#include <type_traits>

#include "tg_dense.h"
#include "tg_sample.h"

namespace tg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4m __attribute__((ext_vector_type(4)));

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

// diagnostic only (TG_GEMM_DBG=16): per-block s_memtime stamps {entry, loop start, loop end, exit}
__device__ unsigned long long g_gemm_trace[4096 * 4];
// The stamps of the LDS-free kernels (k_gemm_ks16, k_gemm_direct, k_gru_direct16) are compiled in with -DTG_PHASE_TRACE only:
// in the production build they cost k_gemm_direct<11, 2, 2> its second block per CU (188 -> 256 registers).
#ifdef TG_PHASE_TRACE
#define TG_PT(...) __VA_ARGS__
#else
#define TG_PT(...)
#endif
// TG_GEMM_DBG=16 stamps every product of a step into the same slots; TG_PHASE_NK=n,k keeps the stamps of the products with
// that output width and inner length only (C2: fc1 172,1204; fc2 172,172; query rows 1032,172) - tools/phase_budget.py
static bool phase_selected(const GemmArgs& g) {
  static const char* sel = getenv("TG_PHASE_NK");
  if (!sel) return true;
  int n = 0, k = 0;
  return sscanf(sel, "%d,%d", &n, &k) == 2 ? (g.n == n && g.k == k) : true;
}

__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16& acc, float bias, float bias2,
                                              int64_t m0, int64_t M, int n0, int wm, int wn, int fr, int fk, int bz);

// One output tile over the k-tiles [kt_begin, kt_end).  raw == nullptr: the normal epilogue; otherwise the
// accumulators are stored unmodified as a row-major [BM][BN] piece for a later fixed-order sum (stream-K
// partials, below).  ASK: the A operand is not read from a0 / a1 but assembled from such pieces while it
// is staged: A[m, k] = act(alpha * (sum of the pieces of element (m, k) + bias[k] + valid[m] * bias2[k])).
template <int WM, int WN, int KS, int D, bool ASK = false, int AP = 3>  // AP: most pieces one element is summed from
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int64_t mt, int nt, int bz, int kt_begin, int kt_end,
                                          float* __restrict__ raw, int trace_slot) {
  const unsigned long long t_entry = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  // KS = 2: a second group of four wavefronts takes the other half of every tile's k-steps into its
  // own accumulators (summed through LDS at the end).  For the under-filled launches of the C2 shapes
  // (a few hundred 64x64 tiles on 1024 SIMDs) this doubles the wavefronts in flight and halves each
  // one's serial MFMA / load chain; large launches keep KS = 1.
  constexpr int THREADS = 256 * KS;
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int RP = THREADS / 8;       // tile rows staged per pass (8 threads per 32-float row)
  constexpr int NA = BM / RP;           // A float4 per thread per tile
  constexpr int NB = BN / RP;           // W float4 per thread per tile (row-major and k-major alike)
  constexpr int NOPS = NA + NB;
  constexpr int PP = 8 / KS;            // k-step pairs per wave per tile
  static_assert(BM % RP == 0 && BN % RP == 0 && NOPS <= PP / 2, "one memory op per k-step pair in each half");
  __shared__ float As[2][BM][LDK];
  __shared__ float Bs[2][BN][LDK];
  __shared__ float red[KS == 2 ? 4 : 1][KS == 2 ? 16 : 1][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, ks = tid >> 8;
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  const int64_t m0 = mt * BM;
  if (m0 >= M) return;
  const int n0 = nt * BN;
  const float* a0p = g.a0.p + (int64_t)bz * g.a0_bs;
  const float* wp = g.w + (int64_t)bz * g.w_bs;
  const int K = g.k, N = g.n, kw0 = g.a0.w;

  const int ar = tid >> 3, ac4 = (tid & 7) * 4;  // A (and row-major W) tile coordinates
  // Branch-free staging: every load is issued unconditionally from a clamped (valid) address,
  // so the compiler can keep two tiles in flight behind counted vmcnt waits.  Rows past M and
  // weight rows past N only feed outputs that are never stored; only k >= K needs zeros.
  const float* arow0[NA];
  const float* arow1[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int64_t m = min(m0 + ar + i * RP, M - 1);
    arow0[i] = a0p + (g.a0.idx ? g.a0.idx[m] : m) * g.a0.ld;
    arow1[i] = g.a1.p ? g.a1.p + (g.a1.idx ? g.a1.idx[m] : m) * g.a1.ld - kw0 : arow0[i];
  }
  int amloc[NA];       // ASK: row inside the producer's 64-row tile, validity byte of the row
  uint8_t avalid[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int64_t m = min(m0 + ar + i * RP, M - 1);
    amloc[i] = (int)(m & 63);
    avalid[i] = (ASK && g.ask_valid) ? g.ask_valid[m] : 0;
  }
  const float* wrow[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (!g.w_kmajor) {
      wrow[i] = wp + (int64_t)min(n0 + ar + i * RP, N - 1) * g.ldw;
    } else {
      const int f = tid + i * THREADS;
      wrow[i] = wp + min(n0 + (f % (BN / 4)) * 4, N - 4);  // column offset; the k row is added per tile
    }
  }
  // D tiles in flight (global -> registers): with the short K of the attention shapes (6-17 tiles) the
  // kernel is a chain of dependent tile loads, so the prefetch distance sets its duration
  static_assert(D >= 2 && D % 2 == 0, "even prefetch depth");
  float4 ra[D][NA], rb[D][NB];
  float4 rx[D][ASK ? NA : 1][AP + 1];  // ASK: pieces 1 .. AP-1, bias, second bias of every staged A float4
  const int nkt = kt_end;  // tiles past the end are clamped to the last one of the range
  // ASK: the pieces of element column k of this block's row tile (producer tile T = mt * NT_p + k / 64)
  auto ask_pieces = [&](int kc, int* wsel, int* psel, bool* have) {
    // (units are dealt per XCD: row tile mt belongs to XCD mt % 8, whose workers are blockIdx = 8 j + mt % 8)
    const int U = g.ask_U;
    const int ua = (((int)mt >> 3) * g.ask_NT + (kc >> 6)) * g.ask_nkt, ub = ua + g.ask_nkt;
    int j0 = (int)((float)ua * g.ask_rcpU);  // ua / U without the integer-division sequence (ua < 2^20: exact after the fix-up)
    j0 += ((j0 + 1) * U <= ua) ? 1 : 0;
    j0 -= (j0 * U > ua) ? 1 : 0;
#pragma unroll
    for (int jj = 0; jj < AP; ++jj) {
      const int j = j0 + jj;
      have[jj] = j * U < ub;
      const int jc = have[jj] ? j : j0;
      wsel[jj] = jc * 8 + ((int)mt & 7);
      psel[jj] = jc * U < ua ? 1 : 0;  // a worker that started in the previous tile holds this one second
    }
  };
  // i-th staged float4 of tile kt (i < NA: A, else W): raw load from a clamped address; columns
  // past K are zeroed when the tile is written to LDS, so nothing waits on the load here
  auto load_one = [&](int kt, int i, float4* ra, float4* rb, float4 (*rx)[AP + 1]) {
    const int k = kt * BK + ac4;
    const int kc = k < K ? k : 0;
    if (ASK && i < NA) {
      int wsel[AP], psel[AP];
      bool have[AP];
      ask_pieces(kc, wsel, psel, have);
      const size_t off = (size_t)amloc[i] * 64 + (kc & 63);
      ra[i] = ldg4(g.ask_part + ((size_t)wsel[0] * 2 + psel[0]) * 4096 + off);
#pragma unroll
      for (int pc = 1; pc < AP; ++pc) rx[i][pc - 1] = ldg4(g.ask_part + ((size_t)wsel[pc] * 2 + psel[pc]) * 4096 + off);
      rx[i][AP - 1] = ldg4(g.ask_bias + kc);
      rx[i][AP] = ldg4((g.ask_bias2 ? g.ask_bias2 : g.ask_bias) + kc);
    } else if (i < NA) {
      ra[i] = ldg4((kc < kw0 ? arow0[i] : arow1[i]) + kc);
    } else if (!g.w_kmajor) {
      rb[i - NA] = ldg4(wrow[i - NA] + kc);
    } else {
      const int kk = kt * BK + (tid + (i - NA) * THREADS) / (BN / 4);
      rb[i - NA] = ldg4(wrow[i - NA] + (int64_t)min(kk, K - 1) * g.ldw);
    }
  };
  auto store_one = [&](int buf, int kt, int i, const float4* ra, const float4* rb, const float4 (*rx)[AP + 1]) {
    const bool kin = kt * BK + ac4 < K;
    if (ASK && i < NA) {
      int wsel[AP], psel[AP];
      bool have[AP];
      ask_pieces(kin ? kt * BK + ac4 : 0, wsel, psel, have);
      const bool b2 = g.ask_bias2 && avalid[i];
      float4 v = ra[i];
#pragma unroll
      for (int pc = 1; pc < AP; ++pc) {  // in worker (= k) order: a fixed order
        if (have[pc]) {
          v.x += rx[i][pc - 1].x; v.y += rx[i][pc - 1].y; v.z += rx[i][pc - 1].z; v.w += rx[i][pc - 1].w;
        }
      }
      auto fin = [&](float p, float b, float c) {
        float o = g.ask_alpha * (p + (b2 ? b + c : b));
        if (g.ask_relu) o = fmaxf(o, 0.f);
        return kin ? o : 0.f;
      };
      sts4(As[buf][ar + i * RP], ac4,
           make_float4(fin(v.x, rx[i][AP - 1].x, rx[i][AP].x), fin(v.y, rx[i][AP - 1].y, rx[i][AP].y),
                       fin(v.z, rx[i][AP - 1].z, rx[i][AP].z), fin(v.w, rx[i][AP - 1].w, rx[i][AP].w)));
    } else if (i < NA) {
      sts4(As[buf][ar + i * RP], ac4, kin ? ra[i] : zero4());
    } else if (!g.w_kmajor) {
      sts4(Bs[buf][ar + (i - NA) * RP], ac4, kin ? rb[i - NA] : zero4());
    } else {
      const int f = tid + (i - NA) * THREADS;
      const int kk = f / (BN / 4), nn = (f % (BN / 4)) * 4;
      const float4 v = (kt * BK + kk < K) ? rb[i - NA] : zero4();
      Bs[buf][nn][kk] = v.x;
      Bs[buf][nn + 1][kk] = v.y;
      Bs[buf][nn + 2][kk] = v.z;
      Bs[buf][nn + 3][kk] = v.w;
    }
  };
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 31, fk = lane >> 5;
  // the bias is requested before the main loop: in the epilogue its load latency (~1 us) would be
  // fully exposed, a third of the block's lifetime at K = 172
  const int n_out = min(n0 + wn * 32 + fr, N - 1);
  const float bias = g.bias ? g.bias[(int64_t)bz * g.bias_bs + n_out] : 0.f;
  const float bias2 = g.bias2 ? g.bias2[n_out] : 0.f;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // Hand-scheduled tile step (same idea as k_gru): the CU's vector-memory path and the
  // in-order wave make bursts of loads / ds_writes stall the matrix pipe, so tile t+2's
  // global loads and tile t+1's LDS writes are threaded between the MFMAs of tile t, with
  // the operand fragments read one k-step pair ahead.  One barrier per tile.
  auto tile = [&](int buf, int kt, float4* la, float4* lb, float4 (*lx)[AP + 1], const float4* sa, const float4* sb,
                  const float4 (*sx)[AP + 1]) {
    const int tl = min(kt + D, nkt - 1);
    const float* ap = &As[buf][wm * 32 + fr][fk + 4 * PP * ks];  // this wave group's share of the k-steps
    const float* bp = &Bs[buf][wn * 32 + fr][fk + 4 * PP * ks];
    float a0 = ap[0], a1 = ap[2], b0 = bp[0], b1 = bp[2];
#pragma unroll
    for (int pr = 0; pr < PP; ++pr) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
      if (pr < PP - 1) {
        na0 = ap[4 * pr + 4]; na1 = ap[4 * pr + 6];
        nb0 = bp[4 * pr + 4]; nb1 = bp[4 * pr + 6];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
      if ((pr % (PP / 2)) < NOPS) {
        if (pr < PP / 2) load_one(tl, pr % (PP / 2), la, lb, lx);
        else store_one(buf ^ 1, kt + 1, pr % (PP / 2), sa, sb, sx);
      }
      __builtin_amdgcn_sched_barrier(0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    __syncthreads();
  };
#pragma unroll
  for (int j = 0; j < D; ++j)
#pragma unroll
    for (int i = 0; i < NOPS; ++i) load_one(min(kt_begin + j, nkt - 1), i, ra[j], rb[j], rx[j]);
#pragma unroll
  for (int i = 0; i < NOPS; ++i) store_one(0, kt_begin, i, ra[0], rb[0], rx[0]);
  __syncthreads();
  const unsigned long long t_loop0 = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  // tile t multiplies LDS[t & 1]; meanwhile tile t + D is loaded into the register slot tile t just
  // left (t % D) and tile t + 1 moves from its slot to the other LDS buffer
  // The loop body is straight-line (whole groups of D tiles; the remainder follows it): with a conditional
  // tile inside the loop the compiler's wait-count bookkeeping loses track of which loads are pending at the
  // joins and waits for ALL of them (vmcnt(0)) at the top of every tile, which throws the prefetch away.
  int kt = kt_begin;
  for (; kt + D <= nkt; kt += D) {
#pragma unroll
    for (int j = 0; j < D; ++j)
      tile(j & 1, kt + j, ra[j], rb[j], rx[j], ra[(j + 1) % D], rb[(j + 1) % D], rx[(j + 1) % D]);
  }
#pragma unroll
  for (int j = 0; j < D - 1; ++j)
    if (kt + j < nkt) tile(j & 1, kt + j, ra[j], rb[j], rx[j], ra[(j + 1) % D], rb[(j + 1) % D], rx[(j + 1) % D]);
  const unsigned long long t_loop1 = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  if (KS == 2) {  // fold the second k-group's partial sums into the first
    if (ks == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    }
    __syncthreads();
    if (ks == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[wave][r][lane];
  }
  if (raw) {  // partial sums of a split tile, row-major [BM][BN]: a wave store covers two full 128-byte rows
#pragma unroll
    for (int r = 0; r < 16; ++r)
      raw[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk) * BN + wn * 32 + fr] = acc[r];
    return;
  }
  gemm_epilogue(g, acc, bias, bias2, m0, M, n0, wm, wn, fr, fk, bz);
  if ((g.dbg & 16) && tid == 0 && trace_slot >= 0 && trace_slot < 4096) {
    g_gemm_trace[trace_slot * 4 + 0] = t_entry;
    g_gemm_trace[trace_slot * 4 + 1] = t_loop0;
    g_gemm_trace[trace_slot * 4 + 2] = t_loop1;
    g_gemm_trace[trace_slot * 4 + 3] = __builtin_amdgcn_s_memtime();
  }
}

// the epilogue of one wave's 32x32 tile: bias / scale / activation / optional per-row operands / store
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16& acc, float bias, float bias2,
                                                   int64_t m0, int64_t M, int n0, int wm, int wn, int fr, int fk, int bz) {
  const int N = g.n;
  const int n = n0 + wn * 32 + fr;
  if (n >= N) return;
  float* cp = g.c + (int64_t)bz * g.c_bs;
  const bool plain = !g.bias_rs && !g.bias2 && !g.row_valid && !g.relu_mask && !g.c_rows && !g.accumulate;
  if (plain) {
    // second destination (write-back rider: h(t-) of the winning positions -> left memory): the 16 row numbers are
    // requested together; wave tiles past the last listed row skip all of it
    const bool two = g.c2 && m0 + wm * 32 < g.c2_m;  // wave-uniform
    int c2r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
      c2r[r] = two ? g.c2_rows[min(m, g.c2_m - 1)] : -1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
      if (m >= M) continue;
      float v = g.alpha * (acc[r] + bias);
      if (g.relu) v = fmaxf(v, 0.f);
      cp[m * g.ldc + n] = v;
      if (two && m < g.c2_m && c2r[r] >= 0) g.c2[(int64_t)c2r[r] * g.ldc2 + n] = v;
    }
  } else {
    // Optional per-row / per-element operands (row scale of the bias, validity bytes, ReLU mask, output
    // row, old value for C +=).  Phase 1 issues EVERY load of the 16 outputs unconditionally - an absent
    // operand reads a harmless valid address (the output tile) and is ignored in phase 2 - so the wave
    // waits for memory once.  With a branch per operand and output the compiler waits after each load:
    // up to 16 x 5 serialised memory latencies per wave (this was a third of the merged fc1 product).
    const float* rsp = g.bias_rs ? g.bias_rs : cp;
    const uint8_t* v2p = g.bias2 ? g.bias2_valid : reinterpret_cast<const uint8_t*>(cp);
    const uint8_t* rvp = g.row_valid ? g.row_valid : reinterpret_cast<const uint8_t*>(cp);
    const float* rmp = g.relu_mask ? g.relu_mask : cp;
    const int* crp = g.c_rows ? g.c_rows : reinterpret_cast<const int*>(cp);
    const int64_t rs_ld = g.bias_rs ? g.ld_brs : 0, rs_o = g.bias_rs ? bz : 0;
    const int64_t rm_ld = g.relu_mask ? g.ld_mask : 0, rm_o = g.relu_mask ? n : 0;
    // four outputs at a time: one exposed latency per group, 24 live registers instead of 96
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float rs[4], rm[4], old[4];
      int crow[4];
      uint8_t v2[4], rv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = q * 4 + j;
        const int64_t m = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, M - 1);
        rs[j] = rsp[m * rs_ld + rs_o];
        v2[j] = v2p[m];
        rv[j] = rvp[m];
        rm[j] = rmp[m * rm_ld + rm_o];
        crow[j] = crp[m];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = q * 4 + j;
        const int64_t m = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, M - 1);
        if (!g.c_rows) crow[j] = (int)m;
        old[j] = cp[(int64_t)crow[j] * g.ldc + n];  // read even without C +=: a valid address either way
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = q * 4 + j;
        const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (m >= M) continue;
        float be = g.bias_rs ? bias * rs[j] : bias;
        if (g.bias2 && v2[j]) be += bias2;
        float v = g.alpha * (acc[r] + be);
        if (g.relu) v = fmaxf(v, 0.f);
        if (g.row_valid && !rv[j]) v = 0.f;
        if (g.relu_mask && !(rm[j] > 0.f)) v = 0.f;
        if (g.accumulate) v += old[j];
        cp[(int64_t)crow[j] * g.ldc + n] = v;
      }
    }
  }
}

extern "C" int tg_debug_gemm_trace(unsigned long long* out_host, int n_blocks) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 4 * n_blocks) == hipSuccess ? 0 : -4;
}

template <int WM, int WN, int KS, int D, bool ASK = false, int AP = 3>
__device__ __forceinline__ void gemm_block(const GemmArgs& g, unsigned bid) {
  constexpr int BN = 32 * WN;
  const int NT = (g.n + BN - 1) / BN;
  const int per = NT * g.nbatch;
  const int xcd = bid & 7, s = bid >> 3;
  const int64_t mt = (int64_t)(s / per) * 8 + xcd;
  const int rem = s % per;
  gemm_tile<WM, WN, KS, D, ASK, AP>(g, mt, rem % NT, rem / NT, 0, (g.k + BK - 1) / BK, nullptr, (int)bid);
}
template <int WM, int WN, int KS, int D, bool ASK = false, int AP = 3>
__global__ void __launch_bounds__(256 * KS) k_gemm(GemmArgs g) {
  gemm_block<WM, WN, KS, D, ASK, AP>(g, blockIdx.x);
}
// ... with a rider as the first r.blocks workgroups (WbRider: tg_common.h; CollateRider: tg_sample.h)
template <class R, int WM, int WN, int KS, int D, bool ASK = false, int AP = 3>
__global__ void __launch_bounds__(256 * KS) k_gemm_r(GemmArgs g, R r) {
  const unsigned own = gridDim.x - r.blocks;  // the product's blocks: before the riders (r.last) or after them
  if (r.last ? blockIdx.x >= own : blockIdx.x < r.blocks) {
    r.run(r.last ? blockIdx.x - own : blockIdx.x);
    return;
  }
  gemm_block<WM, WN, KS, D, ASK, AP>(g, r.last ? blockIdx.x : blockIdx.x - r.blocks);
}

// ---- register-blocked tiles for plain products ----------------------------------------------------------------------
// At C2 sizes the 64 x 64 blocks above are bound by what a CU can pull through its vector-memory path, not by the
// matrix pipe: the G product moves 306 KB per CU for 19.6 k cycles of MFMA work and takes 46 k cycles, and an LDS-free
// form with the SAME bytes per MFMA takes as long (tools/experiments/gemm_direct_no_lds.patch).  Here a wavefront owns
// RM x RN accumulator tiles of 32 x 32, the block (64 RM) x (64 RN): every staged float feeds 2 RN (A) or 2 RM (B) MFMA
// rows instead of two, i.e. 128 x 128 blocks stage half the bytes per MFMA, and a fragment read from LDS feeds RN (RM)
// MFMAs instead of one.  Same pipeline as gemm_tile (two register sets, two LDS buffers, one barrier per k-tile, the
// loads of tile t + 2 and the LDS writes of tile t + 1 threaded between the MFMAs of tile t); plain epilogue only.
template <int RM, int RN>
__device__ __forceinline__ void gemm_rb_block(const GemmArgs& g, unsigned bid) {
  constexpr int BM = 64 * RM, BN = 64 * RN;
  constexpr int RP = 32;                 // tile rows staged per pass (8 threads per 32-float row)
  constexpr int NA = BM / RP, NB = BN / RP, NOPS = NA + NB;
  static_assert(NOPS <= 8, "one memory op per k-step in each half of the tile");
  __shared__ float As[2][BM][LDK];
  __shared__ float Bs[2][BN][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fk = lane >> 5;
  const int K = g.k, N = g.n;
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  const int NT = (N + BN - 1) / BN;
  const int xcd = bid & 7, s = bid >> 3;
  const int64_t mt = (int64_t)(s / NT) * 8 + xcd;
  const int nt = s % NT;
  const int64_t m0 = mt * BM;
  if (m0 >= M) return;
  const int n0 = nt * BN;
  const int ar = tid >> 3, ac4 = (tid & 7) * 4;
  const int kw0 = g.a0.w;  // A = [a0 | a1]: columns [0, kw0) from a0, the rest from a1
  const float* arow[NA];
  const float* arow1[NA];
  const float* wrow[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int64_t m = min(m0 + ar + i * RP, M - 1);
    arow[i] = g.a0.p + (g.a0.idx ? g.a0.idx[m] : m) * g.a0.ld;
    arow1[i] = g.a1.p ? g.a1.p + (g.a1.idx ? g.a1.idx[m] : m) * g.a1.ld - kw0 : arow[i];
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) wrow[i] = g.w + (int64_t)min(n0 + ar + i * RP, N - 1) * g.ldw;
  float bias[RN], bias2[RN];
#pragma unroll
  for (int v = 0; v < RN; ++v) {
    const int nb = min(n0 + (wn * RN + v) * 32 + fr, N - 1);
    bias[v] = g.bias ? g.bias[nb] : 0.f;
    bias2[v] = g.bias2 ? g.bias2[nb] : 0.f;
  }
  const int nkt = (K + BK - 1) / BK;
  float4 ra[2][NA], rb[2][NB];
  auto load_one = [&](int kt, int i, float4* qa, float4* qb) {
    const int k = min(kt, nkt - 1) * BK + ac4;
    const int kc = k < K ? k : 0;
    if (i < NA) qa[i] = ldg4((kc < kw0 ? arow[i] : arow1[i]) + kc);
    else qb[i - NA] = ldg4(wrow[i - NA] + kc);
  };
  auto store_one = [&](int buf, int kt, int i, const float4* qa, const float4* qb) {
    const bool kin = kt * BK + ac4 < K;
    if (i < NA) sts4(As[buf][ar + i * RP], ac4, kin ? qa[i] : zero4());
    else sts4(Bs[buf][ar + (i - NA) * RP], ac4, kin ? qb[i - NA] : zero4());
  };
  f32x16 acc[RM][RN];
#pragma unroll
  for (int u = 0; u < RM; ++u)
#pragma unroll
    for (int v = 0; v < RN; ++v)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[u][v][i] = 0.f;
  auto tile = [&](int buf, int kt, float4* la, float4* lb, const float4* sa, const float4* sb) {
    float a[RM], b[RN], na[RM], nb[RN];
#pragma unroll
    for (int u = 0; u < RM; ++u) a[u] = As[buf][(wm * RM + u) * 32 + fr][fk];
#pragma unroll
    for (int v = 0; v < RN; ++v) b[v] = Bs[buf][(wn * RN + v) * 32 + fr][fk];
#pragma unroll
    for (int st = 0; st < 16; ++st) {  // k-steps of two
      if (st < 15) {
#pragma unroll
        for (int u = 0; u < RM; ++u) na[u] = As[buf][(wm * RM + u) * 32 + fr][fk + 2 * st + 2];
#pragma unroll
        for (int v = 0; v < RN; ++v) nb[v] = Bs[buf][(wn * RN + v) * 32 + fr][fk + 2 * st + 2];
      }
#pragma unroll
      for (int u = 0; u < RM; ++u)
#pragma unroll
        for (int v = 0; v < RN; ++v) {
          acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[v], acc[u][v], 0, 0, 0);
          if (u == 0 && v == 0) {  // this k-step's memory op rides behind its first MFMA
            if (st < 8) { if (st < NOPS) load_one(kt + 2, st, la, lb); }
            else if (st - 8 < NOPS) store_one(buf ^ 1, kt + 1, st - 8, sa, sb);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int u = 0; u < RM; ++u) a[u] = na[u];
#pragma unroll
      for (int v = 0; v < RN; ++v) b[v] = nb[v];
    }
    __syncthreads();
  };
#pragma unroll
  for (int i = 0; i < NOPS; ++i) load_one(0, i, ra[0], rb[0]);
#pragma unroll
  for (int i = 0; i < NOPS; ++i) load_one(1, i, ra[1], rb[1]);
#pragma unroll
  for (int i = 0; i < NOPS; ++i) store_one(0, 0, i, ra[0], rb[0]);
  __syncthreads();
  int kt = 0;
  for (; kt + 2 <= nkt; kt += 2) {  // even tile: LDS[0], loads tile t + 2 -> set 0, stores tile t + 1 (set 1) -> LDS[1]
    tile(0, kt, ra[0], rb[0], ra[1], rb[1]);
    tile(1, kt + 1, ra[1], rb[1], ra[0], rb[0]);
  }
  if (kt < nkt) tile(0, kt, ra[0], rb[0], ra[1], rb[1]);
#pragma unroll
  for (int u = 0; u < RM; ++u)
#pragma unroll
    for (int v = 0; v < RN; ++v) {
      const int n = n0 + (wn * RN + v) * 32 + fr;
      if (n >= N) continue;
      uint8_t v2[16];  // the second bias is added on rows whose validity byte is set; all 16 bytes requested together
      int crow[16];    // ... and the output rows (c_rows: scattered)
      int c2r[16];     // ... and the rows of the second destination (c2: the write-back rider's STEP 6)
      const bool two = g.c2 && m0 + (wm * RM + u) * 32 < g.c2_m;  // wave-uniform
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = min(m0 + (wm * RM + u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, M - 1);
        v2[r] = g.bias2 ? g.bias2_valid[m] : 0;
        crow[r] = g.c_rows ? g.c_rows[m] : (int)m;
        c2r[r] = two ? g.c2_rows[min(m, g.c2_m - 1)] : -1;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + (wm * RM + u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (m >= M) continue;
        float x = g.alpha * (acc[u][v][r] + bias[v] + (v2[r] ? bias2[v] : 0.f));
        if (g.relu) x = fmaxf(x, 0.f);
        g.c[(int64_t)crow[r] * g.ldc + n] = x;
        if (two && m < g.c2_m && c2r[r] >= 0) g.c2[(int64_t)c2r[r] * g.ldc2 + n] = x;
      }
    }
}
template <int RM, int RN>
__global__ void __launch_bounds__(256) k_gemm_rb(GemmArgs g) {
  gemm_rb_block<RM, RN>(g, blockIdx.x);
}
// ---- activation-stationary blocks for short-K, wide-N products ------------------------------------------------------
// The G product of the fused attention (K = d, N = n_head (2d + d_e)) has NKT = 4 .. 8 k-tiles per output tile and many
// column tiles per row tile.  As separate 64 x 64 blocks every column tile re-stages the same 64 x K activation panel
// and pays its own prologue (first loads exposed) and epilogue; at C2 that is 816 blocks of 18 k cycles, a third of it
// outside the k-loop.  Here a block stages its activation panel ONCE (64 x K, odd row stride: conflict-free fragment
// reads) and walks `cpb` column tiles, streaming only weight tiles through the two-buffer pipeline: the stream never
// drains between column tiles (the next tile's first weight tile is already in LDS when a tile's outputs are stored),
// and per k-tile half as much is staged.  NKT is a template parameter so that the tile walk is straight-line.
template <int NKT>
__device__ __forceinline__ void gemm_astat_block(const GemmArgs& g, int cpb, unsigned bid) {
  static_assert(NKT % 2 == 0, "the LDS buffer of a weight tile follows from its k-tile index");
  constexpr int SA = NKT * BK + 1;  // panel row stride (odd)
  __shared__ float As[64][SA];
  __shared__ float Bs[2][64][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fk = lane >> 5;
  const int K = g.k, N = g.n;
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  const int NT = (N + 63) / 64, NG = (NT + cpb - 1) / cpb;  // column tiles, column groups
  const int xcd = bid & 7, s = bid >> 3;
  const int64_t mt = (int64_t)(s / NG) * 8 + xcd;
  const int c0 = (s % NG) * cpb, c1 = min(c0 + cpb, NT);
  const int64_t m0 = mt * 64;
  if (m0 >= M) return;
  const int ar = tid >> 3, ac4 = (tid & 7) * 4;
  int crow[16];  // output rows of this lane's 16 accumulator rows (c_rows: scattered; requested now, used after the loop)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, M - 1);
    crow[r] = g.c_rows ? g.c_rows[m] : (int)m;
  }
  // ---- the activation panel, once
  {
    float4 pa[2][NKT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = min(m0 + ar + 32 * i, M - 1);
      const float* row = g.a0.p + (g.a0.idx ? g.a0.idx[m] : m) * g.a0.ld;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int k = kt * BK + ac4;
        pa[i][kt] = ldg4(row + (k < K ? k : 0));
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) sts4(As[ar + 32 * i], kt * BK + ac4, kt * BK + ac4 < K ? pa[i][kt] : zero4());
  }
  // ---- weight tiles: (column tile c, k-tile j), two register sets / two LDS buffers by the parity of j
  float4 rb[2][2];
  auto load_w = [&](int c, int j, int i, float4* q) {  // i-th staged float4 (rows ar, ar + 32 of the tile)
    const int cc = min(c, c1 - 1);  // past the last column tile: a redundant reload keeps the walk branch-free
    const int k = j * BK + ac4;
    q[i] = ldg4(g.w + (int64_t)min(cc * 64 + ar + 32 * i, N - 1) * g.ldw + (k < K ? k : 0));
  };
  auto store_w = [&](int buf, int j, int i, const float4* q) {
    sts4(Bs[buf][ar + 32 * i], ac4, j * BK + ac4 < K ? q[i] : zero4());
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  load_w(c0, 0, 0, rb[0]); load_w(c0, 0, 1, rb[0]);
  load_w(c0, 1, 0, rb[1]); load_w(c0, 1, 1, rb[1]);
  store_w(0, 0, 0, rb[0]); store_w(0, 0, 1, rb[0]);
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    const int n_out = min(c * 64 + wn * 32 + fr, N - 1);
    const float bias = g.bias ? g.bias[n_out] : 0.f;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {  // straight-line walk over the k-tiles of column tile c
      constexpr int PP = 8;
      const int buf = j & 1;
      // tile (c, j) multiplies; tile (c, j + 2) - or (c + 1, j + 2 - NKT) - is requested into the set tile (c, j) came
      // from; tile (c, j + 1) - or (c + 1, 0) - moves from its set to the other LDS buffer
      const int lc = j + 2 < NKT ? c : c + 1, lj = (j + 2) % NKT, sj = (j + 1) % NKT;
      const float* ap = &As[wm * 32 + fr][j * BK + fk];
      const float* bp = &Bs[buf][wn * 32 + fr][fk];
      float a0 = ap[0], a1 = ap[2], b0 = bp[0], b1 = bp[2];
#pragma unroll
      for (int pr = 0; pr < PP; ++pr) {
        float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
        if (pr < PP - 1) {
          na0 = ap[4 * pr + 4]; na1 = ap[4 * pr + 6];
          nb0 = bp[4 * pr + 4]; nb1 = bp[4 * pr + 6];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
        if (pr < 2) load_w(lc, lj, pr, rb[j & 1]);
        else if (pr >= 4 && pr < 6) store_w(buf ^ 1, sj, pr - 4, rb[(j + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      }
      __syncthreads();
    }
    // outputs of column tile c (plain epilogue); the next tile's first weight tile is already staged
    const int n = c * 64 + wn * 32 + fr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
      float x = g.alpha * (acc[r] + bias);
      if (g.relu) x = fmaxf(x, 0.f);
      if (n < N && m < M) g.c[(int64_t)crow[r] * g.ldc + n] = x;
      acc[r] = 0.f;
    }
  }
}
// ... and for LARGE launches of such products (C5 shape: the query rows of 81 k positive nodes, K = 256 -> N = 1 024; fc2,
// K = N = 256 over 197 k rows), where the register-blocked 128 x 64 blocks re-stage their activation tile for every
// column tile and drain between tiles: eight wavefronts (4 row x 2 column) around a 128-row panel that stays in LDS
// (131 KB at K = 256) while `cpb` column tiles of weights stream through the two-buffer pipeline - 8 KB staged per
// k-tile instead of 24 KB, one float4 per thread.  One block per CU, two wavefronts per SIMD.
template <int NKT>
__global__ void __launch_bounds__(512) k_gemm_astat8(GemmArgs g, int cpb) {
  static_assert(NKT % 2 == 0, "the LDS buffer of a weight tile follows from its k-tile index");
  constexpr int SA = NKT * BK + 1;  // panel row stride (odd)
  __shared__ float As[128][SA];
  __shared__ float Bs[2][64][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fk = lane >> 5;
  const int K = g.k, N = g.n;
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  const int NT = (N + 63) / 64, NG = (NT + cpb - 1) / cpb;
  const int xcd = blockIdx.x & 7, s_ = blockIdx.x >> 3;
  const int64_t mt = (int64_t)(s_ / NG) * 8 + xcd;
  const int c0 = (s_ % NG) * cpb, c1 = min(c0 + cpb, NT);
  const int64_t m0 = mt * 128;
  if (m0 >= M) return;
  const int ar = tid >> 3, ac4 = (tid & 7) * 4;  // 64 rows x 8 float4 per pass
  int crow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, M - 1);
    crow[r] = g.c_rows ? g.c_rows[m] : (int)m;
  }
  {  // the activation panel, once
    float4 pa[2][NKT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = min(m0 + ar + 64 * i, M - 1);
      const float* row = g.a0.p + (g.a0.idx ? g.a0.idx[m] : m) * g.a0.ld;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int k = kt * BK + ac4;
        pa[i][kt] = ldg4(row + (k < K ? k : 0));
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) sts4(As[ar + 64 * i], kt * BK + ac4, kt * BK + ac4 < K ? pa[i][kt] : zero4());
  }
  float4 rb[2];  // weight tiles in flight: two register sets / two LDS buffers by the parity of the k-tile
  auto load_w = [&](int c, int j, float4& q) {
    const int cc = min(c, c1 - 1);
    const int k = j * BK + ac4;
    q = ldg4(g.w + (int64_t)min(cc * 64 + ar, N - 1) * g.ldw + (k < K ? k : 0));
  };
  auto store_w = [&](int buf, int j, const float4& q) { sts4(Bs[buf][ar], ac4, j * BK + ac4 < K ? q : zero4()); };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  load_w(c0, 0, rb[0]);
  load_w(c0, 1, rb[1]);
  store_w(0, 0, rb[0]);
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    const int n_out = min(c * 64 + wn * 32 + fr, N - 1);
    const float bias = g.bias ? g.bias[n_out] : 0.f;
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      constexpr int PP = 8;
      const int buf = j & 1;
      const int lc = j + 2 < NKT ? c : c + 1, lj = (j + 2) % NKT, sj = (j + 1) % NKT;
      const float* ap = &As[wm * 32 + fr][j * BK + fk];
      const float* bp = &Bs[buf][wn * 32 + fr][fk];
      float a0 = ap[0], a1 = ap[2], b0 = bp[0], b1 = bp[2];
#pragma unroll
      for (int pr = 0; pr < PP; ++pr) {
        float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
        if (pr < PP - 1) {
          na0 = ap[4 * pr + 4]; na1 = ap[4 * pr + 6];
          nb0 = bp[4 * pr + 4]; nb1 = bp[4 * pr + 6];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
        if (pr == 0) load_w(lc, lj, rb[j & 1]);
        else if (pr == 4) store_w(buf ^ 1, sj, rb[(j + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      }
      __syncthreads();
    }
    const int n = c * 64 + wn * 32 + fr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
      float x = g.alpha * (acc[r] + bias);
      if (g.relu) x = fmaxf(x, 0.f);
      if (n < N && m < M) g.c[(int64_t)crow[r] * g.ldc + n] = x;
      acc[r] = 0.f;
    }
  }
}

// ---- weight-stationary, LDS-free blocks for short-K products with MANY rows ----------------------------------------------
// C5 shape: fc2 (196 608 x 256 -> 256) and the query rows (81 000 x 256 -> 1 024).  Every LDS-staged form pays a barrier per
// 32-k tile - 16 .. 32 MFMAs per wavefront between barriers - and reached 0.59 / 0.70 of the matrix peak there.  With K this
// short a wavefront can keep its WEIGHTS in registers for the whole launch: lane (i, kq) of a 16 x 16 x 4 MFMA holds the
// 16-byte chunks kq, kq + 4, .. of weight row n0 + i for CW column sets (gemm_direct_tile's operand form: K = 256, 32 columns
// = 128 registers) and walks row tiles of 16: the tile's activation chunks sit in a ring of NS registers, and the moment
// the 4 CW MFMAs of slot s have consumed chunk s of this tile the same register receives chunk s of the NEXT tile - a load
// has a whole tile's MFMAs (128 at K = 256, ~4 000 cycles) to arrive.  No LDS, no barrier, one load per eight MFMAs.
// A block is four wavefronts = four adjacent 32-column groups over the SAME row tiles (their activation loads meet in the
// CU's cache); persistent grid of two blocks per CU; the blocks of one row lane sit on one XCD (blockIdx % 8).
// Plain epilogue (bias, alpha, ReLU, scattered rows), gathered A rows.
template <int NS, bool CROWS, bool IDX>
__global__ void __launch_bounds__(256, 2) k_gemm_wstat(GemmArgs g) {
  constexpr int CW = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int K = g.k, N = g.n;
  const int nch = K / 4;
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  if (M <= 0) return;
  const int NG = (N + 127) / 128;                       // column groups of a block (4 wavefronts x 32 columns)
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int cg = seq % NG;
  const int64_t lane_row = xcd + 8 * (int64_t)(seq / NG);   // this block's row lane
  const int64_t n_lanes = 8 * (int64_t)((gridDim.x >> 3) / NG);
  if (seq / NG >= (int)((gridDim.x >> 3) / NG)) return;     // (grid not a multiple of 8 NG: the surplus blocks idle)
  const int n0 = cg * 128 + wave * 32;
  if (n0 >= N) return;
  const int64_t MT = (M + 15) / 16;
  auto kc = [&](int s_) { return 4 * min(lk + 4 * s_, nch - 1); };  // element offset of this lane's chunk in slot s (clamped into the row)
  // ---- the weights, once (a clamped chunk past K repeats the last one: zeroed, it must not count twice)
  float4 w[CW][NS];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const float* row = g.w + (int64_t)min(n0 + 16 * c + li, N - 1) * g.ldw;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const float4 v = ldg4(row + kc(s_));
      w[c][s_] = (lk + 4 * s_ < nch) ? v : zero4();
    }
  }
  float bias[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) bias[c] = g.bias ? g.bias[min(n0 + 16 * c + li, N - 1)] : 0.f;
  auto arow_id = [&](int64_t mt) {  // the row of A this lane reads in row tile mt (IDX: gathered - a load)
    const int64_t m = min(mt * 16 + li, M - 1);
    if constexpr (IDX) return (int64_t)g.a0.idx[m];
    else return m;
  };
  auto arow = [&](int64_t mt) { return g.a0.p + arow_id(mt) * g.a0.ld; };
  int64_t mt = lane_row;
  if (mt >= MT) return;
  // ring of RS chunk registers: chunk s lives in a[s % RS]; once slot s is multiplied its register receives chunk s + RS - of
  // this tile, or of the next one (K = 256: half a tile = 64 MFMAs ahead; a full-tile ring spills at 256 registers)
  constexpr int RS = (NS > 8 && NS % 2 == 0) ? NS / 2 : NS;  // (the ring is consistent only when RS divides NS)
  float4 a[RS];
  const float* crow_p = arow(mt);
#pragma unroll
  for (int s_ = 0; s_ < RS; ++s_) a[s_] = ldg4(crow_p + kc(s_));
  // One tile: 4 CW NS MFMAs with the next tile's chunk loads threaded between them, then the stores.  The loop body must
  // hold NO branch around a memory instruction (a join makes the wait-count pass wait for everything in flight - the ring
  // lives on loads that stay in flight across a whole tile): full tiles store unconditionally (FULL), the ragged last row
  // tile / column group takes the predicated copy of the body.
  auto tile = [&](auto full_tag, int64_t mt_, int64_t mt_next) {
    constexpr bool FULL = decltype(full_tag)::value;
    const float* const trow = crow_p;  // this tile's rows (the second half of its chunks is still to be requested)
    // the next tile's row id and this tile's output rows are requested FIRST: loads complete in order, so waiting for
    // them later must not mean waiting for the chunk loads that follow
    const int64_t nid = arow_id(mt_next);
    int orow_[4];
    if constexpr (CROWS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) orow_[q] = g.c_rows[min(mt_ * 16 + 4 * lk + q, M - 1)];
    }
    const float* nrow = nullptr;
    f32x4m acc[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[c] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const float4 x = a[s_ % RS];
      const float av[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          const float wv = j == 0 ? w[c][s_].x : j == 1 ? w[c][s_].y : j == 2 ? w[c][s_].z : w[c][s_].w;
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv, acc[c], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      // chunk s + RS into the register this slot has just been multiplied from
      if (s_ + RS == NS || (RS == NS && s_ == 0)) nrow = g.a0.p + nid * g.a0.ld;
      a[s_ % RS] = (s_ + RS < NS) ? ldg4(trow + kc(s_ + RS)) : ldg4(nrow + kc(s_ + RS - NS));
      __builtin_amdgcn_sched_barrier(0);
    }
    crow_p = nrow;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t m = mt_ * 16 + 4 * lk + q;
      int64_t orow = m;
      if constexpr (CROWS) orow = orow_[q];
      float* dst = g.c + orow * g.ldc + n0 + li;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        float x = g.alpha * (acc[c][q] + bias[c]);
        if (g.relu) x = fmaxf(x, 0.f);
        if (FULL) dst[16 * c] = x;
        else if (n0 + 16 * c + li < N && m < M) dst[16 * c] = x;
      }
    }
  };
  using Full = std::integral_constant<bool, true>;
  using Ragged = std::integral_constant<bool, false>;
  const bool cols_full = n0 + 32 <= N;  // wave-uniform
  if (cols_full) {
    for (; (mt + 1) * 16 <= M; mt += n_lanes) {
      const int64_t mt_next = mt + n_lanes < MT ? mt + n_lanes : mt;  // past the end: a redundant reload of this tile
      tile(Full{}, mt, mt_next);
    }
  }
  for (; mt < MT; mt += n_lanes) {
    const int64_t mt_next = mt + n_lanes < MT ? mt + n_lanes : mt;
    tile(Ragged{}, mt, mt_next);
  }
}

template <int NKT>
__global__ void __launch_bounds__(256) k_gemm_astat(GemmArgs g, int cpb) {
  gemm_astat_block<NKT>(g, cpb, blockIdx.x);
}
template <class R, int NKT>
__global__ void __launch_bounds__(256) k_gemm_astat_r(GemmArgs g, int cpb, R r) {
  const unsigned own = gridDim.x - r.blocks;  // the product's blocks: before the riders (r.last) or after them
  if (r.last ? blockIdx.x >= own : blockIdx.x < r.blocks) {
    r.run(r.last ? blockIdx.x - own : blockIdx.x);
    return;
  }
  gemm_astat_block<NKT>(g, cpb, r.last ? blockIdx.x : blockIdx.x - r.blocks);
}

// ---- LDS-free blocks for short-K products with few rows ---------------------------------------------------------------
// The query-row product of the eager step (G_v = c_v Wqk^T + gconst for the ~1 000 positive nodes of a C2 batch: K = d,
// N = n_head (2d + d_e)) is 0.37 GFLOP - 3 us of the matrix pipe - but took 12.8 us as activation-stationary blocks: panel
// through LDS, twelve weight tiles through the two-buffer pipeline, a barrier per tile.  With K this short a wavefront can
// hold its WHOLE operands in registers: lane (i, kq) of a 16 x 16 x 4 MFMA loads the 16-byte chunks kq, kq + 4, kq + 8, ..
// of row i - every chunk of a row exactly once over the four lane quarters - for each of its RW row sets and CW weight-row
// sets, all loads issued back to back (ONE exposed memory latency per block), then NS x 4 x RW x CW MFMAs (step (s, j)
// multiplies element j of chunk slot s on both operands: the sum over k does not care which k values share a step).  No
// LDS, no barrier.  Four wavefronts per block own 2 x 2 wave tiles; a row tile's blocks run on one XCD (chunks of the tile
// sequence, as k_gru_direct16); the grid is 256 persistent blocks sized for the live rows.  Plain epilogue (bias, alpha,
// ReLU, scattered rows).
template <int NS, int RW, int CW>
__device__ __forceinline__ void gemm_direct_tile(const GemmArgs& g, int64_t M, int64_t m0, int n0) {
  TG_PT(const unsigned long long pt_entry = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;)  // (diagnostic, as in gemm_ks16_tile)
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int K = g.k, N = g.n;
  const int nch = K / 4;  // 16-byte chunks per row
  float4 a[RW][NS], w[CW][NS];
  unsigned livem = 0u;    // bit s: chunk slot s of this lane quarter lies inside K
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_)
    if (lk + 4 * s_ < nch) livem |= 1u << s_;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int64_t m = min(m0 + 16 * r + li, M - 1);
    const float* row = g.a0.p + (g.a0.idx ? g.a0.idx[m] : m) * g.a0.ld;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) a[r][s_] = ldg4(row + 4 * min(lk + 4 * s_, nch - 1));
  }
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const float* row = g.w + (int64_t)min(n0 + 16 * c + li, N - 1) * g.ldw;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) w[c][s_] = ldg4(row + 4 * min(lk + 4 * s_, nch - 1));
  }
  float bias[CW];
  int crow[RW][4], c2r[RW][4];
  const bool two = g.c2 && m0 < g.c2_m;  // second destination (write-back rider: STEP 6's rows), wave-uniform
  f32x4m acc[RW][CW];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[r][c] = f32x4m{0.f, 0.f, 0.f, 0.f};
  TG_PT(const unsigned long long pt_loop0 = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;)
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) {
    if (s_ == NS / 2) {
      // epilogue operands (bias of this lane's column in each column set, output rows): requested here, behind half of the
      // MFMAs - early enough to arrive before the epilogue, late enough that the operand registers of the slots already
      // consumed are free (requested with the operands they push the kernel past 256 registers: one block per CU)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CW; ++c) bias[c] = g.bias ? g.bias[min(n0 + 16 * c + li, N - 1)] : 0.f;
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t m = min(m0 + 16 * r + 4 * lk + q, M - 1);
          crow[r][q] = g.c_rows ? g.c_rows[m] : (int)m;
          c2r[r][q] = two ? g.c2_rows[min(m, g.c2_m - 1)] : -1;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool lv = (livem >> s_) & 1u;
    float av[RW][4], wv[CW][4];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float4 x = lv ? a[r][s_] : zero4();  // (a clamped chunk past K repeats the last one: it must not count twice)
      av[r][0] = x.x; av[r][1] = x.y; av[r][2] = x.z; av[r][3] = x.w;
    }
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      wv[c][0] = w[c][s_].x; wv[c][1] = w[c][s_].y; wv[c][2] = w[c][s_].z; wv[c][3] = w[c][s_].w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < CW; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][j], wv[c][j], acc[r][c], 0, 0, 0);
  }
  TG_PT(unsigned long long pt_loop1 = 0ull; if (g.dbg & 16) {
    asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
    pt_loop1 = __builtin_amdgcn_s_memtime();
  })
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int n = n0 + 16 * c + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t m = m0 + 16 * r + 4 * lk + q;
        float x = g.alpha * (acc[r][c][q] + bias[c]);
        if (g.relu) x = fmaxf(x, 0.f);
        if (n < N && m < M) {
          g.c[(int64_t)crow[r][q] * g.ldc + n] = x;
          if (two && m < g.c2_m && c2r[r][q] >= 0) g.c2[(int64_t)c2r[r][q] * g.ldc2 + n] = x;
        }
      }
    }
  TG_PT(if ((g.dbg & 16) && threadIdx.x == 0 && blockIdx.x < 4096) {
    g_gemm_trace[blockIdx.x * 4 + 0] = pt_entry;
    g_gemm_trace[blockIdx.x * 4 + 1] = pt_loop0;
    g_gemm_trace[blockIdx.x * 4 + 2] = pt_loop1;
    g_gemm_trace[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  })
}

// block = 2 x 2 wavefronts of (16 RW) x (16 CW) wave tiles; `own` persistent blocks (riders, if any, sit behind them)
template <int NS, int RW, int CW>
__device__ __forceinline__ void gemm_direct_block(const GemmArgs& g, unsigned bid, unsigned own) {
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  if (M <= 0) return;
  constexpr int BM = 32 * RW, BN = 32 * CW;
  const int NT = (g.n + BN - 1) / BN;
  const int64_t total = ((M + BM - 1) / BM) * NT;
  const int64_t per = (total + 7) / 8;  // XCD x works through the chunk [x per, (x + 1) per) of the tile sequence
  const int wave = threadIdx.x >> 6;
  for (int64_t jx = bid >> 3; jx < per; jx += own >> 3) {
    const int64_t b = (int64_t)(bid & 7) * per + jx;
    if (b >= total) break;
    const int64_t mt = b / NT;
    const int nt = (int)(b - mt * NT);
    const int64_t m0 = mt * BM + (wave >> 1) * (16 * RW);
    const int n0 = nt * BN + (wave & 1) * (16 * CW);
    if (m0 < M && n0 < g.n) gemm_direct_tile<NS, RW, CW>(g, M, m0, n0);
  }
}
// ---- the tail of the split updater (tg_dense.h: GruTail) ----------------------------------------------------------------
// gemm_direct_tile's scheme with RW = 1 and the four column sets being the four PLANES of one 16-column tile: rows of
// [W2 ; W_hh W2] at n0 + li + {0, d, 2d, 3d}.  A wavefront owns 16 rows x 16 hidden columns: 4 NS x 4 MFMAs after ONE
// exposed memory latency, then the gates (update_modules.py:33-37 = torch.nn.GRUCell) with gi read from the buffer the
// second problem of fc1's launch left.  The h plane is the same k-ordered chain as fc2's row of that position.
template <int NS, bool DIRECT>
__device__ __forceinline__ void gru_tail_tile(const GruTail& g, int64_t M, int64_t m0, int n0) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int d = g.d;
  const int nch = d / 4;  // 16-byte chunks per row
  // Two passes of two planes each: all four planes' operands at once are 4 NS + NS float4 = 220 registers at d = 172 - one
  // wavefront per SIMD; here the operands of planes 2, 3 are requested, slot by slot, into the registers planes 0, 1 have
  // just been multiplied from (their latency hides behind the first pass's MFMAs): 3 NS float4, two wavefronts per SIMD.
  // DIRECT: planes r, z then n alone over weight_hh as stored; h is read, not computed
  constexpr int P0 = DIRECT ? 0 : 1;  // weight plane of accumulator 1 (r)
  float4 a[NS], w[2][NS];
  unsigned livem = 0u;
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_)
    if (lk + 4 * s_ < nch) livem |= 1u << s_;
  {
    const int64_t m = min(m0 + li, M - 1);
    const float* row = g.t + g.t_rows[m] * (int64_t)d;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) a[s_] = ldg4(row + 4 * min(lk + 4 * s_, nch - 1));
  }
  const int jc = min(n0 + li, d - 1);
  const float* wrow = g.w + (int64_t)jc * d;
  const int64_t ps = (int64_t)d * d;  // plane stride
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) w[p][s_] = ldg4(wrow + (P0 + p) * ps + 4 * min(lk + 4 * s_, nch - 1));
  float bias[4], gi[4][3], addv[4], hold[4];
  int orow[4];
  f32x4m acc[4];  // r, z | n, h
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    constexpr int NP1 = DIRECT ? 1 : 2;  // planes of the second pass
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      if (pass == 1 && s_ == NS / 2) {  // epilogue operands behind three quarters of the MFMAs (see gemm_direct_tile)
        __builtin_amdgcn_sched_barrier(0);
        if (DIRECT) {
          bias[3] = 0.f;
#pragma unroll
          for (int p = 0; p < 3; ++p) bias[p] = g.b[p * d + jc];
        } else {
          bias[3] = g.b[jc];  // h
#pragma unroll
          for (int p = 0; p < 3; ++p) bias[p] = g.b[(1 + p) * d + jc];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t m = min(m0 + 4 * lk + q, M - 1);
          orow[q] = g.out_rows[m];
#pragma unroll
          for (int p = 0; p < 3; ++p) gi[q][p] = g.gi[m * 3 * (int64_t)d + p * d + jc];
          hold[q] = DIRECT ? g.t[g.t_rows[m] * (int64_t)d + jc] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) addv[q] = (g.out2 && g.add2) ? g.add2[(int64_t)orow[q] * d + jc] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
      }
      const bool lv = (livem >> s_) & 1u;
      const float4 x = lv ? a[s_] : zero4();
      const float av[4] = {x.x, x.y, x.z, x.w};
      float wv[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        wv[p][0] = w[p][s_].x; wv[p][1] = w[p][s_].y; wv[p][2] = w[p][s_].z; wv[p][3] = w[p][s_].w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < (pass == 0 ? 2 : NP1); ++p)
          acc[2 * pass + p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[p][j], acc[2 * pass + p], 0, 0, 0);
      if (pass == 0) {  // this slot's registers are free: the same slot of the remaining planes (n; h = plane 0 of the blob)
        __builtin_amdgcn_sched_barrier(0);
        w[0][s_] = ldg4(wrow + (P0 + 2) * ps + 4 * min(lk + 4 * s_, nch - 1));
        if (!DIRECT) w[1][s_] = ldg4(wrow + 4 * min(lk + 4 * s_, nch - 1));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const int n = n0 + li;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t m = m0 + 4 * lk + q;
    const float h = DIRECT ? hold[q] : acc[3][q] + bias[3];
    const float rg = fast_sigmoid(gi[q][0] + (acc[0][q] + bias[0]));
    const float zg = fast_sigmoid(gi[q][1] + (acc[1][q] + bias[1]));
    const float ng = fast_tanh(gi[q][2] + rg * (acc[2][q] + bias[2]));
    const float hv = (1.f - zg) * ng + zg * h;
    if (n < d && m < M) {
      g.out[(int64_t)orow[q] * d + n] = hv;
      if (g.out2) g.out2[(int64_t)orow[q] * d + n] = hv + addv[q];
    }
  }
}
// block `bid` of g.blocks (a multiple of 8): four wavefronts = four 16-row tiles of one 16-column tile (the weight slab of
// the tile is shared through the CU's cache); XCD chunks of the (row group, column tile) sequence
template <int NS>
__device__ void GruTail::run(unsigned bid) const {
  int64_t M = cap;
  if (n_dev) M = min(M, (int64_t)*n_dev);
  if (M <= 0) return;
  const int NT = (d + 15) / 16;
  const int64_t total = ((M + 63) / 64) * NT;
  const int64_t per = (total + 7) / 8;
  const int wave = threadIdx.x >> 6;
  for (int64_t jx = bid >> 3; jx < per; jx += blocks >> 3) {
    const int64_t b = (int64_t)(bid & 7) * per + jx;
    if (b >= total) break;
    const int64_t mt = b / NT;
    const int nt = (int)(b - mt * NT);
    const int64_t m0 = mt * 64 + 16 * wave;
    if (m0 >= M) continue;
    if (direct) gru_tail_tile<NS, true>(*this, M, m0, nt * 16);
    else gru_tail_tile<NS, false>(*this, M, m0, nt * 16);
  }
}
template <int NS>
__global__ void __launch_bounds__(256) k_gru_tail(GruTail g) {
  g.run<NS>(blockIdx.x);
}
// (the launch of its own always reads h directly: an instance without the pre-multiplied form's fourth plane)
template <int NS>
__global__ void __launch_bounds__(256) k_gru_tail_direct(GruTail g) {
  int64_t M = g.cap;
  if (g.n_dev) M = min(M, (int64_t)*g.n_dev);
  if (M <= 0) return;
  const int NT = (g.d + 15) / 16;
  const int64_t total = ((M + 63) / 64) * NT;
  const int64_t per = (total + 7) / 8;
  const int wave = threadIdx.x >> 6;
  for (int64_t jx = blockIdx.x >> 3; jx < per; jx += gridDim.x >> 3) {
    const int64_t b = (int64_t)(blockIdx.x & 7) * per + jx;
    if (b >= total) break;
    const int64_t mt = b / NT;
    const int nt = (int)(b - mt * NT);
    const int64_t m0 = mt * 64 + 16 * wave;
    if (m0 < M) gru_tail_tile<NS, true>(g, M, m0, nt * 16);
  }
}
static unsigned gru_tail_blocks(const GruTail& t) {
  const int64_t rows = t.rows_hint > 0 ? std::min(t.rows_hint, t.cap) : t.cap;
  return (unsigned)std::min<int64_t>(256, 8 * cdiv(cdiv(rows, 64) * cdiv(t.d, 16), 8));
}
int gru_tail_launch(const GruTail& t, hipStream_t st) {
  if (t.cap <= 0) return TG_OK;
  const int nsl = (int)cdiv(cdiv(t.d, 4), 4);
  if ((t.d % 4) || nsl > 11) return TG_EUNSUPPORTED;
  GruTail g = t;
  g.blocks = gru_tail_blocks(t);
  if (g.direct) {
    if (nsl <= 7) TG_KLAUNCH(k_gru_tail_direct<7>, dim3(g.blocks), dim3(256), 0, st, g);
    else TG_KLAUNCH(k_gru_tail_direct<11>, dim3(g.blocks), dim3(256), 0, st, g);
  } else if (nsl <= 7) {
    TG_KLAUNCH(k_gru_tail<7>, dim3(g.blocks), dim3(256), 0, st, g);
  } else {
    TG_KLAUNCH(k_gru_tail<11>, dim3(g.blocks), dim3(256), 0, st, g);
  }
  return check_launch("gru_tail");
}

template <int NS, class R>
__device__ __forceinline__ void run_rider(const R& r, unsigned bid) {
  if constexpr (std::is_same<R, GruTail>::value) r.template run<NS>(bid);
  else r.run(bid);
}
// a long-K product (k_gemm_ks16's 48 x 48 blocks, A of up to four segments) behind a short-K product's blocks: the split
// updater's W_ih msg shares fc2's launch, whose 144 blocks leave 112 CUs idle at C2
template <int RT, int CT, int NW, int NSEG>
__device__ __forceinline__ void gemm_ks16_blocks(const GemmArgs& g, unsigned bid, unsigned nblk, float* sc_raw);
struct GiRider {
  GemmArgs g;
  unsigned blocks;
};
template <int NS, int RW, int CW>
__global__ void __launch_bounds__(256) k_gemm_direct(GemmArgs g) {
  gemm_direct_block<NS, RW, CW>(g, blockIdx.x, gridDim.x);
}
template <class R, int NS, int RW, int CW>
__global__ void __launch_bounds__(256) k_gemm_direct_r(GemmArgs g, R r) {
  const unsigned own = gridDim.x - r.blocks;  // riders behind the product's blocks
  if (blockIdx.x >= own) {
    if constexpr (std::is_same<R, GiRider>::value) {
      __shared__ float sc_raw[4 * 3 * 9 * 64];
      gemm_ks16_blocks<3, 3, 4, 4>(r.g, blockIdx.x - own, r.blocks, sc_raw);
    } else {
      run_rider<NS>(r, blockIdx.x - own);
    }
    return;
  }
  gemm_direct_block<NS, RW, CW>(g, blockIdx.x, own);
}

// ---- LDS-free K-split blocks for long-K products with few tiles ------------------------------------------------------
// The merged value / out / fc1 product of the fused attention (C2: M = 3 072, N = 172, K = 1 204) has 144 tiles of 64 x 64
// for 256 CUs; as stream-K pieces (below) it fills the chip but leaves its result in two or three pieces per element for
// the consumer to sum.  This is k_gru_direct16's scheme with one plane: a block owns (16 RT) x (16 CT) outputs - 48 x 48
// makes 64 x 4 = 256 blocks of exactly that product - the 32-k tiles are dealt to NW wavefronts, lane (i, kq) of a
// 16 x 16 x 4 MFMA feeds the matrix unit from the 32 contiguous bytes A[i][k0 + 8 kq ..] / W[j][k0 + 8 kq ..] it loads
// itself (no LDS in the k-loop; A may be two column segments, rows optionally gathered), the wavefronts' accumulators meet
// in one reduce-scatter round through LDS (summed in wavefront order: bit-reproducible) and the first four wavefronts
// run the epilogue (bias, second bias on valid rows, alpha, ReLU) - the product leaves its final values, so its consumer
// is a plain product.  256 persistent blocks, XCD chunks of the tile sequence.
// NSEG = 4: A is four column segments, every one optionally gathered, the third possibly a slice of zeros (GemmArgs.a2 / a3)
template <int RT, int CT, int NW, int NSEG = 2>
__device__ __forceinline__ void gemm_ks16_tile(const GemmArgs& g, int64_t M, int64_t m0, int n0, float* sc_raw) {
  constexpr int NS = RT * CT;  // 16 x 16 subtiles of the block
  // diagnostic only (g.dbg & 16; tools/phase_budget.py): s_memtime stamps {entry, first tile requested, k-loop done, exit}
  // of the block's first tile, wavefront 0
  TG_PT(const unsigned long long pt_entry = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;)
  float (*sc)[NW - 1][NS][64] = reinterpret_cast<float (*)[NW - 1][NS][64]>(sc_raw);  // [owner 0..3][slot][subtile][lane]
  const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int K = g.k, N = g.n, kw0 = g.a0.w;
  const int kw1 = kw0 + g.a1.w, kw2 = kw1 + (NSEG == 4 ? g.a2.w : 0);  // segment ends (NSEG = 4)
  const bool zero2 = NSEG == 4 && !g.a2.p;                              // the third segment is zeros
  const float* ar0[RT];
  const float* ar1[RT];
  const float* ar2[NSEG == 4 ? RT : 1];
  const float* ar3[NSEG == 4 ? RT : 1];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t m = min(m0 + 16 * rt + li, M - 1);
    ar0[rt] = g.a0.p + (g.a0.idx ? g.a0.idx[m] : m) * g.a0.ld;
    ar1[rt] = g.a1.p ? g.a1.p + (g.a1.idx ? g.a1.idx[m] : m) * g.a1.ld - kw0 : ar0[rt];
    if (NSEG == 4) {
      ar2[rt] = g.a2.p ? g.a2.p + (g.a2.idx ? g.a2.idx[m] : m) * g.a2.ld - kw1 : ar0[rt];
      ar3[rt] = g.a3.p ? g.a3.p + (g.a3.idx ? g.a3.idx[m] : m) * g.a3.ld - kw2 : ar0[rt];
    }
  }
  const float* wr[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) wr[ct] = g.w + (int64_t)min(n0 + 16 * ct + li, N - 1) * g.ldw;
  const int nkt = (K + BK - 1) / BK;
  const int n_my = ks < nkt ? (nkt - ks + NW - 1) / NW : 0;
  struct Tile {
    float4 a[RT][2], w[CT][2];
  };
  // raw loads from clamped addresses (see k_gru_direct), split in two: the addresses of a tile, then its 2 (RT + CT)
  // loads one at a time - threaded between the MFMAs of the tile before (a burst of twelve loads in front of a tile's 72
  // MFMAs holds the matrix pipe for the time it takes to issue them)
  struct Addr {
    int kc[2];
    unsigned live;
  };
  auto tile_addr = [&](int i, Addr& A) {
    const int t = ks + NW * max(0, min(i, n_my - 1));
    const int kb = min(t, nkt - 1) * BK + 8 * lk;
    A.live = 0u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = kb + 4 * q;
      const bool dead = k >= K || (zero2 && k >= kw1 && k < kw2);  // (a chunk of the zero segment: multiplied as zeros)
      if (!dead && ks < nkt) A.live |= 1u << q;
      A.kc[q] = k < K ? k : 0;
    }
  };
  auto load_one = [&](int l, const Addr& A, Tile& T) {  // l = 0 .. 2 (RT + CT) - 1
    const int q = l & 1, r = l >> 1;
    if (r < RT) {
      const int k = A.kc[q];
      const float* row;
      if (NSEG == 4) row = k < kw0 ? ar0[r] : k < kw1 ? ar1[r] : k < kw2 ? ar2[r] : ar3[r];
      else row = k < kw0 ? ar0[r] : ar1[r];
      T.a[r][q] = ldg4(row + k);
    } else {
      T.w[r - RT][q] = ldg4(wr[r - RT] + A.kc[q]);
    }
  };
  f32x4m acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4m{0.f, 0.f, 0.f, 0.f};
  constexpr int NL = 2 * (RT + CT), NM = 8 * RT * CT;  // loads / MFMAs per tile
  // multiply tile T (live mask lv) and, threaded through it, request the next tile (addresses An) into Tn
  auto mma_tile = [&](const Tile& T, unsigned lv, const Addr& An, Tile& Tn, bool prefetch) {
    int mi = 0, li_ = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float av[RT][4], wv[CT][4];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float4 a = ((lv >> q) & 1u) ? T.a[rt][q] : zero4();
        av[rt][0] = a.x; av[rt][1] = a.y; av[rt][2] = a.z; av[rt][3] = a.w;
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        wv[ct][0] = T.w[ct][q].x; wv[ct][1] = T.w[ct][q].y; wv[ct][2] = T.w[ct][q].z; wv[ct][3] = T.w[ct][q].w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][j], wv[ct][j], acc[rt][ct], 0, 0, 0);
            ++mi;
            if (prefetch && li_ < NL && mi * NL >= (li_ + 1) * NM / 2) {  // the loads ride in the first half of the tile
              load_one(li_, An, Tn);
              ++li_;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
    }
  };
  // epilogue operands of the outputs wavefront ks < 4 finishes (accumulator register ks of every subtile: row 4 lk + ks),
  // requested before the loop
  float bias[CT], bias2[CT];
  uint8_t v2[RT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int n = min(n0 + 16 * ct + li, N - 1);
    bias[ct] = g.bias ? g.bias[n] : 0.f;
    bias2[ct] = g.bias2 ? g.bias2[n] : 0.f;
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) v2[rt] = g.bias2 ? g.bias2_valid[min(m0 + 16 * rt + 4 * lk + (ks & 3), M - 1)] : 0;
  // (ablations at C2's fc1 shape, 3 072 x 1 204 -> 172: loads alone 7.2 us, MFMAs alone 16.6 us, both 19.6 us - the block is
  // bound by its four wavefronts' MFMA streams, 720 dependent-free MFMAs of 32 cycles each; a third register set, two
  // tiles ahead, costs the second block per CU and is slower: 20.4 us)
  Tile T0, T1;
  Addr A0, A1;
  tile_addr(0, A0);
#pragma unroll
  for (int l = 0; l < NL; ++l) load_one(l, A0, T0);
  TG_PT(const unsigned long long pt_loop0 = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;)
  int i = 0;
  for (; i + 2 <= n_my; i += 2) {
    tile_addr(i + 1, A1);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(T0, A0.live, A1, T1, true);
    __builtin_amdgcn_sched_barrier(0);
    tile_addr(i + 2, A0);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(T1, A1.live, A0, T0, true);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (i < n_my) mma_tile(T0, A0.live, A1, T1, false);
  TG_PT(unsigned long long pt_loop1 = 0ull; if (g.dbg & 16) {  /* the stamp waits for the last MFMA */
    asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
    pt_loop1 = __builtin_amdgcn_s_memtime();
  })
  // reduce-scatter: wavefront v < 4 finishes accumulator register v of every subtile; everybody parks the registers the
  // others own (one round, one barrier); sums run in wavefront order 0 .. NW - 1
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    if (v != ks) {  // wave-uniform
      const int slot = ks < v ? ks : ks - 1;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) sc[v][slot][rt * CT + ct][lane] = acc[rt][ct][v];
    }
  }
  __syncthreads();
  if (ks >= 4) return;
  float o[RT][CT];
#pragma unroll
  for (int src = 0; src < NW; ++src) {
    float t[RT][CT];
    if (src == ks) {  // wave-uniform
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (v == ks) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) t[rt][ct] = acc[rt][ct][v];
        }
    } else {
      const int slot = src < ks ? src : src - 1;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) t[rt][ct] = sc[ks][slot][rt * CT + ct][lane];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) o[rt][ct] = src == 0 ? t[rt][ct] : o[rt][ct] + t[rt][ct];
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t m = m0 + 16 * rt + 4 * lk + ks;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int n = n0 + 16 * ct + li;
      float x = g.alpha * (o[rt][ct] + bias[ct] + (v2[rt] ? bias2[ct] : 0.f));
      if (g.relu) x = fmaxf(x, 0.f);
      if (n < N && m < M) g.c[m * g.ldc + n] = x;
    }
  }
  TG_PT(if ((g.dbg & 16) && tid == 0 && blockIdx.x < 4096) {
    g_gemm_trace[blockIdx.x * 4 + 0] = pt_entry;
    g_gemm_trace[blockIdx.x * 4 + 1] = pt_loop0;
    g_gemm_trace[blockIdx.x * 4 + 2] = pt_loop1;
    g_gemm_trace[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  })
}

// persistent blocks of one product: block `bid` of `nblk` (a multiple of 8) works through its XCD's chunk of the tile sequence
template <int RT, int CT, int NW, int NSEG>
__device__ __forceinline__ void gemm_ks16_blocks(const GemmArgs& g, unsigned bid, unsigned nblk, float* sc_raw) {
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  if (M <= 0) return;
  constexpr int BM = 16 * RT, BN = 16 * CT;
  const int NT = (g.n + BN - 1) / BN;
  const int64_t total = ((M + BM - 1) / BM) * NT;
  const int64_t per = (total + 7) / 8;
  for (int64_t jx = bid >> 3; jx < per; jx += nblk >> 3) {
    const int64_t b = (int64_t)(bid & 7) * per + jx;
    if (b >= total) break;
    const int64_t mt = b / NT;
    const int nt = (int)(b - mt * NT);
    gemm_ks16_tile<RT, CT, NW, NSEG>(g, M, mt * BM, nt * BN, sc_raw);
    __syncthreads();  // the fold's LDS is re-used by the next tile
  }
}
// A second product in the same launch (S2 = Ks16Second): `blocks` further persistent blocks behind the first product's, in
// front of the riders.  Two resident blocks per CU then interleave their MFMA streams (one block's four wavefronts - one
// per SIMD, issuing in order - keep the matrix pipe ~40 % busy).
struct NoSecond {
  unsigned blocks;
};
struct Ks16Second {
  GemmArgs g;
  unsigned blocks;
  unsigned seq;  // 1: no further blocks - the first product's blocks work through the second product's tiles afterwards
};
template <class R, int RT, int CT, int NW, class S2 = NoSecond>
__global__ void __launch_bounds__(64 * NW) k_gemm_ks16(GemmArgs g, R r, S2 s2) {
  __shared__ float sc_raw[4 * (NW - 1) * RT * CT * 64];
  const unsigned own = gridDim.x - r.blocks - s2.blocks;  // persistent blocks of the product; second product, then riders
  if (blockIdx.x >= own + s2.blocks) {
    r.run(blockIdx.x - own - s2.blocks);
    return;
  }
  if constexpr (std::is_same<S2, Ks16Second>::value) {
    if (blockIdx.x >= own) {
      gemm_ks16_blocks<RT, CT, NW, 4>(s2.g, blockIdx.x - own, s2.blocks, sc_raw);
      return;
    }
  }
  gemm_ks16_blocks<RT, CT, NW, 2>(g, blockIdx.x, own, sc_raw);
  if constexpr (std::is_same<S2, Ks16Second>::value) {
    if (s2.seq) gemm_ks16_blocks<RT, CT, NW, 4>(s2.g, blockIdx.x, own, sc_raw);
  }
}

// ---- fc1 and fc2 of the attention block in ONE launch ------------------------------------------------------------------
// The block's last two products are t = relu([S | c] W1f^T + b1 + valid c1) (K = n_head kvw + d) and h = t W2^T + b2 (K = d).
// A block that owns WHOLE rows of t can run fc2 on them as its epilogue: 16 rows x all d columns (CT column subtiles of 16),
// the k-loop of k_gemm_ks16 (LDS-free, K split over four wavefronts) with 1 x CT subtiles, the fold leaves the tile of t in
// LDS, and the four wavefronts then share fc2's column subtiles: A fragments from that tile, W2 rows straight from memory
// (requested before the fold), 4 NS MFMAs per subtile, h stored - and scattered a second time into the left memory for
// the winning positions (STEP 6, GemmArgs.c2 form).  C2: 192 blocks of 16 rows instead of 256 blocks of 48 x 48 + a launch
// of 144 blocks; fc2's launch - ten microseconds of which one is matrix work - is gone, the 64 CUs the product leaves idle
// host the write-back riders.  Measured slower (see gemm_fc12_launch): kept as an opt-in form.
struct Fc2Fuse {
  const float* w;    // [n2, ldw] (torch Linear layout), n2 <= 16 CT
  int64_t ldw;
  const float* bias;
  float* c;          // [M, ldc]
  int64_t ldc;
  int n;             // output columns of fc2
  float* c2;         // nullable: second, scattered destination (rows c2_rows[m] for m < c2_m, -1 = none)
  const int32_t* c2_rows;
  int64_t c2_m, ldc2;
};
template <int CT>
__device__ __forceinline__ void gemm_ks16_fc2_tile(const GemmArgs& g, const Fc2Fuse& f, int64_t M, int64_t m0, float* t_raw) {
  // COLUMN split: wavefront ks owns column subtiles CW ks .. CW ks + CW - 1 of the 16-row block over the WHOLE K - no fold,
  // ~100 registers (a K split over the four wavefronts with all CT subtiles per wavefront needs two tiles of 2 (1 + CT)
  // operand chunks in flight: 512 registers and spills, 37.5 us at C2 against 33.5 us for the two launches).  The operand
  // chunks of D = 4 k-tiles are in flight: a tile's registers are reloaded for tile i + D right after its MFMAs.
  constexpr int NW = 4, CW = (CT + NW - 1) / NW, D = 4;
  constexpr int TS = 16 * CT + 4;              // row stride of the t tile in LDS
  constexpr int NS2 = (16 * CT + 15) / 16;     // chunk slots of fc2's K (= columns of t, 16 per slot)
  float (*ts)[TS] = reinterpret_cast<float (*)[TS]>(t_raw);
  const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int K = g.k, N = g.n, kw0 = g.a0.w;
  const int64_t m = min(m0 + li, M - 1);
  const float* ar0 = g.a0.p + (g.a0.idx ? g.a0.idx[m] : m) * g.a0.ld;
  const float* ar1 = g.a1.p ? g.a1.p + (g.a1.idx ? g.a1.idx[m] : m) * g.a1.ld - kw0 : ar0;
  const float* wr[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) wr[c] = g.w + (int64_t)min(16 * (CW * ks + c) + li, N - 1) * g.ldw;
  const int nkt = (K + BK - 1) / BK;
  struct Tile {
    float4 a[2], w[CW][2];
  };
  auto load_tile = [&](int t, Tile& T) {  // raw loads from clamped addresses; chunks past K are zeroed when they are used
    const int kb = min(t, nkt - 1) * BK + 8 * lk;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = kb + 4 * q;
      const int kc = k < K ? k : 0;
      T.a[q] = ldg4((kc < kw0 ? ar0 : ar1) + kc);
#pragma unroll
      for (int c = 0; c < CW; ++c) T.w[c][q] = ldg4(wr[c] + kc);
    }
  };
  f32x4m acc[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) acc[c] = f32x4m{0.f, 0.f, 0.f, 0.f};
  auto mma_tile = [&](int t, const Tile& T) {
    const int kb = t * BK + 8 * lk;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 a = (kb + 4 * q < K) ? T.a[q] : zero4();
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          const float wv = j == 0 ? T.w[c][q].x : j == 1 ? T.w[c][q].y : j == 2 ? T.w[c][q].z : T.w[c][q].w;
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv, acc[c], 0, 0, 0);
        }
    }
  };
  Tile T[D];
#pragma unroll
  for (int b = 0; b < D; ++b) load_tile(b, T[b]);
  // epilogue operands of fc1 and fc2's weights: requested behind the first tiles, long before they are needed
  float bias[CW], bias2[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int n = min(16 * (CW * ks + c) + li, N - 1);
    bias[c] = g.bias ? g.bias[n] : 0.f;
    bias2[c] = g.bias2 ? g.bias2[n] : 0.f;
  }
  uint8_t v2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v2[q] = g.bias2 ? g.bias2_valid[min(m0 + 4 * lk + q, M - 1)] : 0;
  int t = 0;
  for (; t + D <= nkt; t += D) {
#pragma unroll
    for (int b = 0; b < D; ++b) {
      mma_tile(t + b, T[b]);
      __builtin_amdgcn_sched_barrier(0);
      load_tile(t + b + D, T[b]);  // (past the end: a redundant reload of the last tile)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int b = 0; b < D; ++b)
    if (t + b < nkt) mma_tile(t + b, T[b]);
  // fc2's operands of this wavefront's column subtiles (the same CW ks .. of fc2's output columns)
  const int K2 = N, nch2 = K2 / 4;
  float4 w2[CW][NS2];
  float b2[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int n2 = min(16 * (CW * ks + c) + li, f.n - 1);
    const float* row = f.w + (int64_t)n2 * f.ldw;
    b2[c] = f.bias ? f.bias[n2] : 0.f;
#pragma unroll
    for (int s_ = 0; s_ < NS2; ++s_) {
      const float4 v = ldg4(row + 4 * min(lk + 4 * s_, nch2 - 1));
      w2[c][s_] = (lk + 4 * s_ < nch2) ? v : zero4();
    }
  }
  int crow2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t mm = min(m0 + 4 * lk + q, M - 1);
    crow2[q] = (f.c2 && m0 < f.c2_m) ? f.c2_rows[min(mm, f.c2_m - 1)] : -1;
  }
  // t = relu(alpha (acc + b1 + valid c1)); the tile goes to LDS (columns past N: zeros - they are fc2's k range too)
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int n = 16 * (CW * ks + c) + li;
    if (n >= 16 * CT) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x = g.alpha * (acc[c][q] + bias[c] + (v2[q] ? bias2[c] : 0.f));
      if (g.relu) x = fmaxf(x, 0.f);
      ts[4 * lk + q][n] = n < N ? x : 0.f;
      if (g.c && n < N && m0 + 4 * lk + q < M) g.c[(m0 + 4 * lk + q) * g.ldc + n] = x;  // (t itself, when somebody reads it)
    }
  }
  __syncthreads();
  // ---- fc2 on the tile: h[16, n2] = t W2^T + b2, this wavefront's column subtiles (their MFMA chains interleaved)
  f32x4m a2[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) a2[c] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s_ = 0; s_ < NS2; ++s_) {
    const int kk = 4 * min(lk + 4 * s_, (16 * CT) / 4 - 1);
    const float4 x = *reinterpret_cast<const float4*>(&ts[li][kk]);
    const bool lv = lk + 4 * s_ < nch2;
    const float av[4] = {lv ? x.x : 0.f, lv ? x.y : 0.f, lv ? x.z : 0.f, lv ? x.w : 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const float wv = j == 0 ? w2[c][s_].x : j == 1 ? w2[c][s_].y : j == 2 ? w2[c][s_].z : w2[c][s_].w;
        a2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv, a2[c], 0, 0, 0);
      }
  }
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int n2 = 16 * (CW * ks + c) + li;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t mm = m0 + 4 * lk + q;
      const float hv = a2[c][q] + b2[c];
      if (n2 < f.n && mm < M) {
        f.c[mm * f.ldc + n2] = hv;
        if (crow2[q] >= 0 && mm < f.c2_m) f.c2[(int64_t)crow2[q] * f.ldc2 + n2] = hv;
      }
    }
  }
}
template <class R, int CT>
__global__ void __launch_bounds__(256) k_gemm_ks16_fc2(GemmArgs g, Fc2Fuse f, R r) {
  __shared__ float t_raw[16 * (16 * CT + 4)];
  const unsigned own = gridDim.x - r.blocks;  // persistent blocks of the product; riders behind them
  if (blockIdx.x >= own) {
    r.run(blockIdx.x - own);
    return;
  }
  int64_t M = g.m_cap;
  if (g.m_dev) M = min(M, (int64_t)*g.m_dev);
  if (M <= 0) return;
  const int64_t total = (M + 15) / 16;
  const int64_t per = (total + 7) / 8;  // XCD x works through the chunk [x per, (x + 1) per) of the row tiles
  for (int64_t jx = blockIdx.x >> 3; jx < per; jx += own >> 3) {
    const int64_t b = (int64_t)(blockIdx.x & 7) * per + jx;
    if (b >= total) break;
    gemm_ks16_fc2_tile<CT>(g, f, M, b * 16, t_raw);
    __syncthreads();  // the LDS tiles are re-used by the next row tile
  }
}

// Is the product one for k_gemm_ks16?  Long K, plain epilogue (+ second bias), few enough 48 x 48 tiles that the blocks
// fit the chip in one round or two.  TG_GEMM_KS16: 0 = off, 8 = eight wavefronts per block.
static unsigned rider_blocks(int64_t live, int threads, int64_t rows);  // (below, with gemm_launch)
static bool ks16_second_ok(const GemmArgs* s);
static unsigned ks16_second_blocks(const GemmArgs& s);
struct NoRider {  // (k_gemm_ks16 without riders)
  unsigned blocks;
  __device__ __forceinline__ void run(unsigned) const {}
};
bool gemm_ks16_launch(const GemmArgs& g, hipStream_t st, const WbRider* rider, bool* rode, const GemmArgs* second,
                      bool* second_rode) {
  static const int knob = getenv("TG_GEMM_KS16") ? atoi(getenv("TG_GEMM_KS16")) : 1;
  if (rode) *rode = false;
  if (second_rode) *second_rode = false;
  if (!knob || g.m_cap <= 0 || g.nbatch != 1 || g.w_kmajor || g.bias_rs || g.row_valid || g.relu_mask || g.c_rows || g.accumulate ||
      g.c2 || g.ask_part || (g.k % 4) || (g.a0.w % 4) || (g.ldw % 4) || g.a0.w + (g.a1.p ? g.a1.w : 0) != g.k)
    return false;
  // column tiles of 48 (CT = 3), or - narrow outputs (N <= 112, e.g. LastFM's --dim 100: 64-column tiles waste 28 % of the
  // MFMAs, 48-column tiles 44 %) - ONE column tile of 16 CT = 64 / 112 columns with 32-row blocks
  const int ct = g.n <= 64 ? 4 : g.n <= 112 ? 7 : 3;
  // (measured, N = 100: 24 576 x 500 42.2 -> 39.6 us, 6 144 x 500 17.7 -> 13.3 us, 600 x 500 16.6 -> 9.2 us; 3 072 x 400 -> 64
  // 14.5 -> 6.4 us; at K = 300 the 64 x 64 blocks are faster: 27.9 against 30.9 us)
  // (a launch of at most one 48 x 48 block per CU pays from K = 320: the score head's product, 2 B x 2 d -> d, at d = 172
  // 2 048 rows 13.3 -> 6.8 us, 400 rows 12.9 -> 4.9 us against the 64 x 64 LDS blocks; 4 096 rows and more: no gain)
  static const int mink_knob = getenv("TG_GEMM_KS16_MINK") ? atoi(getenv("TG_GEMM_KS16_MINK")) : 0;  // tuning knob
  const bool one_round = ct == 3 && cdiv(g.m_cap, 48) * cdiv(g.n, 48) <= 256;
  if (g.k < (mink_knob ? mink_knob : (ct == 3 ? (one_round ? 320 : 512) : 384))) return false;
  const int64_t ntc = cdiv(g.n, 16 * ct);
  static const int64_t max_tiles = getenv("TG_GEMM_KS16_TILES") ? atoi(getenv("TG_GEMM_KS16_TILES")) : 1100;  // tuning knob (measured, K = 1 204, N = 172: 6 144 rows 54.5 -> 37.1 us, 12 288 rows 78.2 -> 72.1 us, 24 576 rows 123 -> 129 us)
  if (cdiv(g.m_cap, ct == 3 ? 48 : 32) * ntc > max_tiles) return false;
  GemmArgs gd = g;
  static const int gdbg16 = getenv("TG_GEMM_DBG") ? (atoi(getenv("TG_GEMM_DBG")) & 16) : 0;  // diagnostic: phase stamps
  gd.dbg = (gdbg16 && phase_selected(g)) ? 16 : 0;
  const NoRider nr{0u};
  const NoSecond ns{0u};
  if (ct != 3) {
    // (rows per block: 16 when that fits the chip at once, else 32)
    const bool r1 = cdiv(g.m_cap, 16) * ntc <= 256;
    if (ct == 4) {
      if (r1) TG_KLAUNCH((k_gemm_ks16<NoRider, 1, 4, 4>), dim3(256), dim3(256), 0, st, gd, nr, ns);
      else TG_KLAUNCH((k_gemm_ks16<NoRider, 2, 4, 4>), dim3(256), dim3(256), 0, st, gd, nr, ns);
    } else {
      if (r1) TG_KLAUNCH((k_gemm_ks16<NoRider, 1, 7, 4>), dim3(256), dim3(256), 0, st, gd, nr, ns);
      else TG_KLAUNCH((k_gemm_ks16<NoRider, 2, 7, 4>), dim3(256), dim3(256), 0, st, gd, nr, ns);
    }
    return true;
  }
  const int64_t nt48 = ntc;
  // rows per block: the smallest of 16 / 32 / 48 whose blocks fit the chip at once (fewer rows = a shorter block)
  const int rt = cdiv(g.m_cap, 16) * nt48 <= 256 ? 1 : cdiv(g.m_cap, 32) * nt48 <= 256 ? 2 : 3;
  if (rider && rode && knob != 8) {
    // the write-back rider (STEP 4-5 depend on nothing the attention block computes) on THIS launch, the longest of the
    // block: its chain of dependent round trips (~9 us) hides behind the product instead of extending fc2's launch
    // (C2: fc1 22.9 -> 25.8 us, fc2 + riders 16.8 -> 12.9 us; with the short blocks of a small batch fc2's launch hides the
    // rider as well and fc1 only gets longer: C1 13.1 -> 14.5 us)
    static const int here = getenv("TG_WB_RIDER_FC1") ? atoi(getenv("TG_WB_RIDER_FC1")) : 1;  // tuning knob: 0 = on fc2
    if (here && rt == 3 && cdiv(g.m_cap, 48) * nt48 <= 256) {
      WbRider wr = *rider;
      const int64_t live = std::min<int64_t>(256, cdiv(g.m_cap, 16 * rt) * nt48);
      wr.blocks = rider_blocks(live, 256, 2 * wr.a.B);
      wr.last = 1u;
      // (second product: plain epilogue, A of up to four segments, K a multiple of 4, final values)
      const bool two = second_rode && ks16_second_ok(second);
      if (two) {
        Ks16Second s2{*second, 0u, 0u};
        s2.g.dbg = 0;
        // TG_KS16_SECOND (tuning knob): 0 = co-resident blocks of the second product, 1 = the same blocks, afterwards
        static const int seq_knob = getenv("TG_KS16_SECOND") ? atoi(getenv("TG_KS16_SECOND")) : 1;
        s2.seq = seq_knob == 1 ? 1u : 0u;
        s2.blocks = s2.seq ? 0u : ks16_second_blocks(*second);
        const dim3 gr2(256 + s2.blocks + wr.blocks);
        TG_KLAUNCH((k_gemm_ks16<WbRider, 3, 3, 4, Ks16Second>), gr2, dim3(256), 0, st, gd, wr, s2);
        *second_rode = true;
      } else {
        const dim3 gr(256 + wr.blocks);
        TG_KLAUNCH((k_gemm_ks16<WbRider, 3, 3, 4>), gr, dim3(256), 0, st, gd, wr, ns);
      }
      *rode = true;
      return true;
    }
  }
  if (knob == 8) TG_KLAUNCH((k_gemm_ks16<NoRider, 3, 3, 8>), dim3(256), dim3(512), 0, st, gd, nr, ns);
  else if (rt == 1) TG_KLAUNCH((k_gemm_ks16<NoRider, 1, 3, 4>), dim3(256), dim3(256), 0, st, gd, nr, ns);
  else if (rt == 2) TG_KLAUNCH((k_gemm_ks16<NoRider, 2, 3, 4>), dim3(256), dim3(256), 0, st, gd, nr, ns);
  else TG_KLAUNCH((k_gemm_ks16<NoRider, 3, 3, 4>), dim3(256), dim3(256), 0, st, gd, nr, ns);
  return true;
}

// fc1 + fc2 in one launch (k_gemm_ks16_fc2): g = fc1 (long K, plain epilogue + second bias, N <= 176), g2 = fc2 over fc1's
// output (K = g.n, plain epilogue, optionally the second scattered destination); one round of 16-row blocks that leaves
// CUs for the riders.  MEASURED at C2 (1x MI355X, 100 replays): parity-green and SLOWER - the launch takes 47 us against
// 23.2 + 10.3 us for the two (step 0.0994 against 0.0850 ms); as a K split over the four wavefronts with all 11 subtiles per
// wavefront (512 registers, spills) 37.5 us.  A block of 16 whole rows reads ALL of W1f (828 KB) for 16 x 176 outputs - 1.6 x
// the operand bytes per flop of the 48 x 48 blocks (159 MB against 118 MB per launch through the CUs' memory paths), and
// that traffic, not the launch count, bounds fc1 at this size.  Opt-in: TG_FC12=1.
bool gemm_fc12_launch(const GemmArgs& g, const GemmArgs& g2, hipStream_t st, const WbRider* rider, bool* rode) {
  static const int knob = getenv("TG_FC12") ? atoi(getenv("TG_FC12")) : 0;  // tuning knob (default off, see above)
  if (rode) *rode = false;
  if (!knob || g.m_cap <= 0 || g.nbatch != 1 || g.w_kmajor || g.bias_rs || g.row_valid || g.relu_mask || g.c_rows || g.accumulate ||
      g.c2 || g.ask_part || (g.k % 4) || (g.a0.w % 4) || (g.ldw % 4) || g.a0.w + (g.a1.p ? g.a1.w : 0) != g.k || g.k < 512)
    return false;
  if (g.n > 176 || g.n <= 112 || (g.n % 4)) return false;  // (narrower outputs: the 48 / 112-column blocks of k_gemm_ks16)
  if (g2.nbatch != 1 || g2.w_kmajor || g2.bias_rs || g2.bias2 || g2.row_valid || g2.relu_mask || g2.c_rows || g2.accumulate ||
      g2.ask_part || g2.a1.p || g2.a0.idx || g2.relu || g2.alpha != 1.f || g2.k != g.n || g2.n > 176 || (g2.ldw % 4) ||
      g2.m_cap != g.m_cap || g2.m_dev != g.m_dev || g2.a0.p != g.c)
    return false;
  const int64_t tiles = cdiv(g.m_cap, 16);
  if (tiles > 232 || tiles < 128) return false;  // one round with CUs to spare; few tiles: the 48-column blocks fill more CUs
  GemmArgs gd = g;
  gd.dbg = 0;
  gd.c = nullptr;  // t itself has no other reader
  Fc2Fuse f{g2.w, g2.ldw, g2.bias, g2.c, g2.ldc, g2.n, g2.c2, g2.c2_rows, g2.c2_m, g2.ldc2};
  const unsigned own = (unsigned)(8 * cdiv(tiles, 8));
  if (rider && rode) {
    WbRider wr = *rider;
    wr.blocks = rider_blocks(own, 256, 2 * wr.a.B);
    wr.last = 1u;
    TG_KLAUNCH((k_gemm_ks16_fc2<WbRider, 11>), dim3(own + wr.blocks), dim3(256), 0, st, gd, f, wr);
    *rode = true;
  } else {
    const NoRider nr{0u};
    TG_KLAUNCH((k_gemm_ks16_fc2<NoRider, 11>), dim3(own), dim3(256), 0, st, gd, f, nr);
  }
  return true;
}

// ---- stream-K for launches that cannot fill the chip --------------------------------------------------
// A product with fewer 64x64 tiles than CUs and a long K (the merged value/out/fc1 product of the fused
// attention: 144 tiles x 38 k-tiles on 256 CUs) leaves CUs idle for its whole duration.  Here the
// (tile, k-tile) units are dealt evenly: worker w owns units [w U, (w+1) U) of the tile-major sequence, i.e.
// the tail of one tile and/or the head of the next, and stores each piece's accumulators in its own slot.
// Nobody waits for anybody: the pieces of a tile are summed, in worker order (a fixed order: deterministic),
// by the CONSUMER of the product while it stages its A operand (gemm_tile<..., ASK>), together with the
// producer's bias / activation.  U < nkt is required (every tile is split); with few tiles a tile is cut into up to
// eight pieces (consumer instance AP = 8), with 128 .. 255 tiles into two or three (AP = 3).
template <int KS, int D>
__global__ void __launch_bounds__(256 * KS) k_gemm_sk(GemmArgs g, SkPlan sk) {
  // Units are dealt per XCD (blockIdx % 8, as the hardware deals blocks): the row tiles mt = x (mod 8) belong to
  // XCD x, as in k_gemm's map, and so do the consumer's blocks of these rows - every piece is written and read
  // inside one XCD's L2 (32 workers x 2 slots x 16 KB = 1 MB of its 4 MB).
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int64_t total = (int64_t)((sk.MT - x + 7) / 8) * sk.NT * sk.nkt;
  int64_t u = (int64_t)j * sk.U;
  const int64_t u1 = min(u + sk.U, total);
  int piece = 0;
  while (u < u1) {  // at most two pieces (U < nkt)
    const int lt = (int)(u / sk.nkt), kt0 = (int)(u % sk.nkt);
    const int kt1 = (int)min((int64_t)sk.nkt, kt0 + (u1 - u));
    gemm_tile<2, 2, KS, D>(g, (int64_t)(lt / sk.NT) * 8 + x, lt % sk.NT, 0, kt0, kt1,
                           sk.part + ((size_t)blockIdx.x * 2 + piece) * 4096, -1);
    __syncthreads();  // the LDS tiles are re-used by the next piece
    u += kt1 - kt0;
    ++piece;
  }
}

bool gemm_sk_partials(const GemmArgs& g, float* ws, size_t ws_floats, hipStream_t st, SkPlan* plan) {
  static const int sk_knob = getenv("TG_GEMM_SK") ? atoi(getenv("TG_GEMM_SK")) : 1;  // tuning knob: 0 = off
  if (!sk_knob || !ws || g.m_cap <= 0 || g.nbatch != 1 || g.m_dev || (g.k % 4) || (g.a0.w % 4) || (g.ldw % 4) ||
      g.a0.w + (g.a1.p ? g.a1.w : 0) != g.k)
    return false;
  SkPlan p{};
  p.NT = (int)cdiv(g.n, 64);
  p.MT = (int)cdiv(g.m_cap, 64);
  p.tiles = p.MT * p.NT;
  p.nkt = (int)cdiv(g.k, BK);
  static const int wk_knob = getenv("TG_SK_WORKERS") ? atoi(getenv("TG_SK_WORKERS")) : TG_SK_WORKERS;  // tuning knob
  const int workers = std::min(std::max(wk_knob & ~7, 8), TG_SK_WORKERS_MAX);
  if (p.tiles >= TG_SK_WORKERS || p.nkt < 16) return false;  // (also covers few tiles: then a tile is cut into more pieces)
  const int64_t units_xcd = cdiv((int64_t)p.MT, 8) * p.NT * p.nkt;  // of the fullest XCD
  p.U = (int)cdiv(units_xcd, (int64_t)(workers / 8));
  p.pieces = (int)cdiv(p.nkt, p.U) + 1;  // most pieces a tile can be cut into (its range may start mid-worker)
  if (p.U >= p.nkt || p.pieces > 8 || p.U < 3 || ws_floats < (size_t)workers * 2 * 4096) return false;
  p.part = ws;
  static const int gdbg = getenv("TG_GEMM_DBG") ? atoi(getenv("TG_GEMM_DBG")) : 0;
  GemmArgs gd = g;
  gd.dbg = gdbg & ~16;
  TG_KLAUNCH((k_gemm_sk<2, 2>), dim3(workers), dim3(512), 0, st, gd, p);
  *plan = p;
  return true;
}

// Riders (WbRider, CollateRider) pay where the product's launch is small: its live blocks leave CUs idle (C2: 144 blocks of
// fc2 + 112 write-back riders, one block per CU either way) or fill the chip for a round or so.  Large launches take no
// rider - measured at C5 shape: the write-back as 512 riders of fc2 costs 240 us where its own launch takes 180 us, and
// the collate riders of the query-row product change nothing - the caller then launches that work itself.  `live`: the
// product's blocks that have rows (a launch may be sized for a capacity several times its live rows: GemmArgs.m_hint).
// Riders sit BEHIND the product's blocks in the grid: the dispatcher starts the product - the critical path - first, the
// riders take what is left.
constexpr int64_t RIDER_MAX_LIVE = 1024;
static int64_t live_blocks(const GemmArgs& g, int64_t grid, int bm) {
  if (g.m_hint <= 0 || g.m_hint >= g.m_cap) return grid;
  return std::max<int64_t>(1, grid * cdiv(g.m_hint, (int64_t)bm) / std::max<int64_t>(1, cdiv(g.m_cap, (int64_t)bm)));
}
// write-back riders: one wavefront per listed row at most; the idle CUs of an under-filled launch, else two per CU; a
// multiple of 8 (the product's XCD map stays)
static unsigned rider_blocks(int64_t live, int threads, int64_t rows) {
  const int64_t want = cdiv(rows, (int64_t)(threads / 64));
  const int64_t room = live <= 248 ? (256 - live) * (threads <= 256 ? 2 : 1) : 512;  // (idle CUs host two 4-wave blocks)
  return (unsigned)std::max<int64_t>(8, std::min(want, room) & ~(int64_t)7);
}
// collate riders: the sampler is a chain of dependent memory round trips per query and wants many wavefronts (its own
// launch has one group of 16 lanes per query); the centres are a gather with four elements per thread in flight
static void collate_blocks(CollateRider& c) {
  const int64_t Q = 3 * c.s.B;
  const int64_t sg = cdiv(Q, (int64_t)16), cg = cdiv(Q * (c.cr.m.d / 4), (int64_t)256);
  c.sblocks = (c.parts == 2) ? 0u : (unsigned)std::min<int64_t>(sg, 1024);
  c.cr.blocks = (c.parts == 1) ? 0u : (unsigned)std::min<int64_t>(cdiv(cg, (int64_t)4), 512);
  c.blocks = (c.sblocks + c.cr.blocks + 7u) & ~7u;
  c.last = 1u;
}

// may the product ride as k_gemm_ks16 blocks (plain epilogue, final values, A of two to four segments)?
static bool ks16_second_ok(const GemmArgs* s) {
  return s && s->m_cap > 0 && s->nbatch == 1 && !s->w_kmajor && !s->bias_rs && !s->row_valid && !s->relu_mask && !s->c_rows &&
         !s->accumulate && !s->c2 && !s->ask_part && !(s->k % 4) && !(s->a0.w % 4) && !(s->a1.w % 4) && !(s->a2.w % 4) &&
         !(s->ldw % 4) && s->a0.w + s->a1.w + s->a2.w + s->a3.w == s->k && s->a0.p && (s->a1.p || !s->a1.w) && (s->a3.p || !s->a3.w);
}
static unsigned ks16_second_blocks(const GemmArgs& s) {
  const int64_t rows2 = s.m_hint > 0 ? std::min(s.m_hint, s.m_cap) : s.m_cap;
  return (unsigned)std::min<int64_t>(256, 8 * cdiv(cdiv(rows2, 48) * cdiv(s.n, 48), 8));
}
int gemm_launch(const GemmArgs& g, hipStream_t st, const WbRider* rider, bool* rode, const CollateRider* collate,
                const GruTail* tail, const GemmArgs* second) {
  if (rode) *rode = false;
  if ((rider || collate || tail || second) &&
      (!rode || ((rider != nullptr) + (collate != nullptr) + (tail != nullptr) + (second != nullptr) > 1)))
    return TG_EINVAL;
  if (g.c2 && (g.bias_rs || g.bias2 || g.row_valid || g.relu_mask || g.c_rows || g.accumulate || !g.c2_rows || g.nbatch != 1))
    return TG_EINVAL;  // the second destination exists in the plain epilogue only
  if (g.m_cap <= 0) return TG_OK;
  if (g.n <= 0 || g.k <= 0 || (g.k % 4) || (g.a0.w % 4) || (g.ldw % 4) || g.nbatch <= 0) return TG_EINVAL;
  if (g.w_kmajor && (g.n % 4)) return TG_EINVAL;
  static const int log_knob = getenv("TG_GEMM_LOG") ? atoi(getenv("TG_GEMM_LOG")) : 0;  // diagnostic: one line per product
  if (log_knob)
    fprintf(stderr, "gemm m_cap=%lld m_dev=%d hint=%lld n=%d k=%d nbatch=%d kmajor=%d a1=%d idx=%d relu=%d mask=%d acc=%d c_rows=%d bias2=%d brs=%d\n",
            (long long)g.m_cap, g.m_dev != nullptr, (long long)g.m_hint, g.n, g.k, g.nbatch, g.w_kmajor, g.a1.p != nullptr,
            g.a0.idx != nullptr, g.relu, g.relu_mask != nullptr, g.accumulate, g.c_rows != nullptr, g.bias2 != nullptr, g.bias_rs != nullptr);
  if (g.a0.w + (g.a1.p ? g.a1.w : 0) != g.k) return TG_EINVAL;
  // long K, few tiles: LDS-free K-split blocks (k_gemm_ks16)
  if (!rider && !collate && !tail && !second && !g.bias2 && gemm_ks16_launch(g, st)) return check_launch("gemm(ks16)");
  constexpr int BM = 64, BN = 64;
  const int64_t MT = cdiv(g.m_cap, BM);
  const int NT = (int)cdiv(g.n, BN);
  const int64_t grid = 8 * cdiv(MT, 8) * NT * g.nbatch;
  static const int ks_knob = getenv("TG_GEMM_KS") ? atoi(getenv("TG_GEMM_KS")) : 0;  // tuning knob: 1 / 2, 0 = auto
  // measured at C2: a second k-group helps only launches that leave CUs idle AND have a long K (the merged
  // value/out/fc1 product: 144 tiles x 38 k-steps, 37 -> 35 us); elsewhere it is neutral or slightly worse
  const bool split = ks_knob ? ks_knob == 2 : (grid <= 256 && g.k >= 512);
  static const int gdbg = getenv("TG_GEMM_DBG") ? atoi(getenv("TG_GEMM_DBG")) : 0;
  GemmArgs gd = g;
  gd.dbg = phase_selected(g) ? gdbg : (gdbg & ~16);
  static const int depth_knob = getenv("TG_GEMM_DEPTH") ? atoi(getenv("TG_GEMM_DEPTH")) : 2;  // tuning knob: 2 / 4
  // a write-back rider that is not hosted: the caller runs the whole write-back itself, STEP 6's rows included - this
  // launch then stores no second copy of them
  auto no_ride = [&]() { if (rider) gd.c2 = nullptr; };
  // plain row-major products with MANY rows: register-blocked 128 x 64 blocks (three quarters of the staged bytes per
  // MFMA, half the fragment reads).  Measured: C5 shape (24 576 / 6 144 / 6 144 blocks) G 1.00 -> 0.96 ms, fc1 1.19 ->
  // 1.08 ms, fc2 0.261 -> 0.253 ms; at C3 / C4 sizes (288 .. 2 500 blocks) 3 .. 17 % SLOWER than the 64 x 64 blocks (whose
  // several co-resident blocks per CU cover each other's prologue and epilogue); 128 x 128 blocks are slower still there
  static const int rb_knob = getenv("TG_GEMM_RB") ? atoi(getenv("TG_GEMM_RB")) : 1;  // tuning knob: 0 = off, 2 = 128 x 128
  // short K (4 / 6 / 8 k-tiles), plain epilogue, MANY rows: panel-stationary blocks of eight wavefronts (k_gemm_astat8)
  // (measured, 81 000 x 256 -> 1 024: 434 us as register-blocked blocks, 396 / 375 / 434 us with 4 / 8 / 16 column tiles per
  // panel block; 196 608 x 256 -> 256: 267 us with 4 against 260-280)
  static const int as8_knob = getenv("TG_GEMM_ASTAT8") ? atoi(getenv("TG_GEMM_ASTAT8")) : 8;  // tuning knob: column tiles per block, 0 = off
  {
    const int nkt8 = (int)cdiv(g.k, BK);
    const bool plain8 = !g.ask_part && g.nbatch == 1 && !g.a1.p && !g.w_kmajor && !g.bias_rs && !g.bias2 && !g.row_valid &&
                        !g.relu_mask && !g.accumulate && !g.c2 && g.a0.w == g.k;
    // short K, plain epilogue, MANY rows: weight-stationary LDS-free blocks (k_gemm_wstat; TG_GEMM_WSTAT=0: off)
    static const int ws_knob = getenv("TG_GEMM_WSTAT") ? atoi(getenv("TG_GEMM_WSTAT")) : 1;  // tuning knob
    // Measured (1x MI355X): 196 608 x 256 -> 256 (fc2 at C5 shape) 265 -> 245 us (105 TF/s); 81 920 x 256 -> 1 024 383 us against
    // 378 us for the panel-stationary blocks below, 140 001 x 172 -> 1 032 536 against 568 us.  Ablation: without the chunk
    // reloads (MFMA stream + epilogue alone) 328 us = 0.83 of the matrix peak - the reloads cost the rest although they
    // are requested half a tile ahead.  Taken for N <= 256 (TG_GEMM_WSTAT=2: every N).
    if (ws_knob && plain8 && g.k <= 256 && g.k >= 64 && (g.n <= 256 || ws_knob == 2) && cdiv(g.m_cap, 128) * NT >= 4096) {
      no_ride();
      const int NG = (int)cdiv(g.n, 128);
      const unsigned lanes8 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(512 / 8 / NG, cdiv(cdiv(g.m_cap, 16), 8)));
      const dim3 gr(8u * (unsigned)NG * lanes8);
      const int nsl = (int)cdiv(cdiv(g.k, 4), 4);
#define TG_WSTAT(NS_)                                                                                   \
  do {                                                                                                  \
    if (g.c_rows && g.a0.idx) TG_KLAUNCH((k_gemm_wstat<NS_, true, true>), gr, dim3(256), 0, st, gd);    \
    else if (g.c_rows) TG_KLAUNCH((k_gemm_wstat<NS_, true, false>), gr, dim3(256), 0, st, gd);          \
    else if (g.a0.idx) TG_KLAUNCH((k_gemm_wstat<NS_, false, true>), gr, dim3(256), 0, st, gd);          \
    else TG_KLAUNCH((k_gemm_wstat<NS_, false, false>), gr, dim3(256), 0, st, gd);                       \
  } while (0)
      if (nsl <= 8) TG_WSTAT(8);
      else if (nsl <= 11) TG_WSTAT(11);
      else TG_WSTAT(16);
#undef TG_WSTAT
      return check_launch("gemm(wstat)");
    }
    if (as8_knob && plain8 && (nkt8 == 4 || nkt8 == 6 || nkt8 == 8) && cdiv(g.m_cap, 128) * NT >= 4096) {
      no_ride();
      const int cpb = std::min(as8_knob, NT);
      const dim3 gr((unsigned)(8 * cdiv(cdiv(g.m_cap, 128), 8) * cdiv(NT, cpb)));
      if (nkt8 == 4) TG_KLAUNCH((k_gemm_astat8<4>), gr, dim3(512), 0, st, gd, cpb);
      else if (nkt8 == 6) TG_KLAUNCH((k_gemm_astat8<6>), gr, dim3(512), 0, st, gd, cpb);
      else TG_KLAUNCH((k_gemm_astat8<8>), gr, dim3(512), 0, st, gd, cpb);
      return check_launch("gemm(astat8)");
    }
  }
  if (rb_knob && !g.ask_part && g.nbatch == 1 && !g.w_kmajor && !g.bias_rs && !g.row_valid && !g.relu_mask &&
      !g.accumulate && cdiv(g.m_cap, 128) * NT >= 4096) {
    no_ride();
    // (128 x 128 blocks - four 32 x 32 accumulators per wavefront, twice the MFMAs between barriers - where N is a multiple
    // of 128 and K long: 196 608 x 1 280 -> 256 1 072 -> 1 041 us; K = 256: no difference)
    if ((rb_knob == 2 && g.n >= 512) || (rb_knob == 1 && g.n % 128 == 0 && g.k >= 512) || (rb_knob == 3 && g.n >= 128)) {
      TG_KLAUNCH((k_gemm_rb<2, 2>), dim3((unsigned)(8 * cdiv(cdiv(g.m_cap, 128), 8) * cdiv(g.n, 128))), dim3(256), 0, st, gd);
    } else {
      TG_KLAUNCH((k_gemm_rb<2, 1>), dim3((unsigned)(8 * cdiv(cdiv(g.m_cap, 128), 8) * NT)), dim3(256), 0, st, gd);
    }
    return check_launch("gemm(rb)");
  }
  // short K and few (live) rows: whole operands in registers (see gemm_direct_tile).  Wave tiles of 32 x 32, or 32 x 48
  // when that brings the blocks of the estimated live rows under the CU count
  static const int dir_knob = getenv("TG_GEMM_DIRECT") ? atoi(getenv("TG_GEMM_DIRECT")) : 1;  // tuning knob: 0 = off
  {
    const int nsl = (int)cdiv(cdiv(g.k, 4), 4);  // chunk slots per lane quarter
    const int64_t rows_e = g.m_hint > 0 ? std::min(g.m_hint, g.m_cap) : g.m_cap;
    const bool plain_d = !g.ask_part && g.nbatch == 1 && !g.a1.p && !g.w_kmajor && !g.bias_rs && !g.bias2 && !g.row_valid &&
                         !g.relu_mask && !g.accumulate && (!g.c2 || !g.c_rows) && g.a0.w == g.k;
    static const int64_t dir_tiles = getenv("TG_GEMM_DIRECT_TILES") ? atoi(getenv("TG_GEMM_DIRECT_TILES")) : 512;  // tuning knob
    if (dir_knob && plain_d && nsl <= 12 && cdiv(rows_e, 64) * cdiv(g.n, 64) <= dir_tiles) {
      const int64_t tiles64 = cdiv(rows_e, 64) * cdiv(g.n, 64);
      const bool wide = tiles64 > 256 && nsl <= 11;  // (32 x 48 wave tiles: 5 x 44 operand registers)
      const unsigned own = 256;
      const bool ride = nsl <= 11;  // (the K <= 192 instance hosts no riders)
      CollateRider co = (collate && ride) ? *collate : CollateRider{};
      WbRider wr = (rider && ride) ? *rider : WbRider{};
      GruTail tl = (tail && ride && (int)cdiv(cdiv(tail->d, 4), 4) == nsl) ? *tail : GruTail{};
      if (tl.cap > 0) tl.blocks = gru_tail_blocks(tl);
      GiRider gi{};
      if (ride && ks16_second_ok(second)) {
        gi.g = *second;
        gi.g.dbg = 0;
        gi.blocks = ks16_second_blocks(*second);
      }
      if (collate && ride) collate_blocks(co);
      if (rider && ride) {
        wr.blocks = rider_blocks(std::min<int64_t>(tiles64, 256), 256, 2 * wr.a.B);
        wr.last = 1u;
      } else {
        no_ride();
      }
      const dim3 gr(own + co.blocks + wr.blocks + tl.blocks + gi.blocks);
#define TG_DIRECT(NS_)                                                                                                  \
  do {                                                                                                                  \
    if (gi.blocks && wide) TG_KLAUNCH((k_gemm_direct_r<GiRider, NS_, 2, 3>), gr, dim3(256), 0, st, gd, gi);       \
    else if (gi.blocks) TG_KLAUNCH((k_gemm_direct_r<GiRider, NS_, 2, 2>), gr, dim3(256), 0, st, gd, gi);         \
    else if (tl.blocks && wide) TG_KLAUNCH((k_gemm_direct_r<GruTail, NS_, 2, 3>), gr, dim3(256), 0, st, gd, tl);  \
    else if (tl.blocks) TG_KLAUNCH((k_gemm_direct_r<GruTail, NS_, 2, 2>), gr, dim3(256), 0, st, gd, tl);         \
    else if (wr.blocks && wide) TG_KLAUNCH((k_gemm_direct_r<WbRider, NS_, 2, 3>), gr, dim3(256), 0, st, gd, wr);      \
    else if (wr.blocks) TG_KLAUNCH((k_gemm_direct_r<WbRider, NS_, 2, 2>), gr, dim3(256), 0, st, gd, wr);        \
    else if (co.blocks && wide) TG_KLAUNCH((k_gemm_direct_r<CollateRider, NS_, 2, 3>), gr, dim3(256), 0, st, gd, co); \
    else if (co.blocks) TG_KLAUNCH((k_gemm_direct_r<CollateRider, NS_, 2, 2>), gr, dim3(256), 0, st, gd, co);   \
    else if (wide) TG_KLAUNCH((k_gemm_direct<NS_, 2, 3>), gr, dim3(256), 0, st, gd);                            \
    else TG_KLAUNCH((k_gemm_direct<NS_, 2, 2>), gr, dim3(256), 0, st, gd);                                      \
  } while (0)
      if (nsl <= 7) TG_DIRECT(7);
      else if (nsl <= 11) TG_DIRECT(11);
      else TG_KLAUNCH((k_gemm_direct<12, 2, 2>), dim3(own), dim3(256), 0, st, gd);
#undef TG_DIRECT
      if (((collate || rider) && ride) || tl.blocks || gi.blocks) *rode = true;
      return check_launch("gemm(direct)");
    }
  }
  // short K, many column tiles, a launch of a few blocks per CU: activation-stationary blocks (see k_gemm_astat)
  static const int as_knob = getenv("TG_GEMM_ASTAT") ? atoi(getenv("TG_GEMM_ASTAT")) : 1;  // tuning knob: 0 = off, n = column tiles per block
  {
    const int nkt = (int)cdiv(g.k, BK);
    const bool plain_as = !g.ask_part && g.nbatch == 1 && !g.a1.p && !g.w_kmajor && !g.bias_rs && !g.bias2 && !g.row_valid &&
                          !g.relu_mask && !g.accumulate && g.a0.w == g.k;
    if (as_knob && plain_as && NT >= 8 && (nkt == 4 || nkt == 6 || nkt == 8) && MT * NT <= 1024) {  // (C3's 3 264 tiles: no gain)
      // two column tiles per block (measured at C2, 48 x 17 tiles: 21.6 us; three: 28.1, four: 24.6, six: 29.8, nine:
      // 40.8; separate 64 x 64 blocks: 24.4): two such blocks share a CU and cover each other's barriers, which matters
      // more than the shared panel
      static const int cpb_knob = getenv("TG_GEMM_ASTAT_CPB") ? atoi(getenv("TG_GEMM_ASTAT_CPB")) : 0;
      const int cpb = cpb_knob > 0 ? std::min(cpb_knob, NT) : as_knob > 1 ? std::min(as_knob, NT) : 2;
      const int64_t gb = 8 * cdiv(MT, 8) * cdiv(NT, cpb);
      if (collate && live_blocks(g, gb, 64) <= RIDER_MAX_LIVE) {
        CollateRider co = *collate;
        collate_blocks(co);
        const dim3 grid_r((unsigned)(gb + co.blocks));
        if (nkt == 4) TG_KLAUNCH((k_gemm_astat_r<CollateRider, 4>), grid_r, dim3(256), 0, st, gd, cpb, co);
        else if (nkt == 6) TG_KLAUNCH((k_gemm_astat_r<CollateRider, 6>), grid_r, dim3(256), 0, st, gd, cpb, co);
        else TG_KLAUNCH((k_gemm_astat_r<CollateRider, 8>), grid_r, dim3(256), 0, st, gd, cpb, co);
        *rode = true;
        return check_launch("gemm(astat+collate)");
      }
      no_ride();
      const dim3 grid_as((unsigned)gb);
      if (nkt == 4) TG_KLAUNCH((k_gemm_astat<4>), grid_as, dim3(256), 0, st, gd, cpb);
      else if (nkt == 6) TG_KLAUNCH((k_gemm_astat<6>), grid_as, dim3(256), 0, st, gd, cpb);
      else TG_KLAUNCH((k_gemm_astat<8>), grid_as, dim3(256), 0, st, gd, cpb);
      return check_launch("gemm(astat)");
    }
  }
  if (g.ask_part) {  // A assembled from stream-K pieces of a 64-row-tiled producer with g.k output columns
    if (g.nbatch != 1 || g.w_kmajor || g.k != g.a0.w || !g.ask_bias || g.ask_NT != (int)cdiv(g.k, 64)) return TG_EINVAL;
    static const int ask_depth = getenv("TG_GEMM_ASK_DEPTH") ? atoi(getenv("TG_GEMM_ASK_DEPTH")) : 2;  // tuning knob: 2 / 4
    gd.ask_rcpU = 1.0f / (float)g.ask_U;
    static const int ask_ks = getenv("TG_GEMM_ASK_KS") ? atoi(getenv("TG_GEMM_ASK_KS")) : 2;  // tuning knob: 1 / 2 (measured 12.6 / 11.1 us)
    WbRider wr = rider ? *rider : WbRider{};
    if (rider && (g.ask_pieces > 3 || ask_ks == 2)) {  // (stream-K consumers: the producer had fewer tiles than CUs)
      wr.blocks = rider_blocks(grid, 512, 2 * wr.a.B);
      wr.last = 1u;
      if (g.ask_pieces > 3)
        TG_KLAUNCH((k_gemm_r<WbRider, 2, 2, 2, 2, true, 8>), dim3((unsigned)grid + wr.blocks), dim3(512), 0, st, gd, wr);
      else
        TG_KLAUNCH((k_gemm_r<WbRider, 2, 2, 2, 2, true>), dim3((unsigned)grid + wr.blocks), dim3(512), 0, st, gd, wr);
      *rode = true;
    } else {
      no_ride();
      if (g.ask_pieces > 3) TG_KLAUNCH((k_gemm<2, 2, 2, 2, true, 8>), dim3((unsigned)grid), dim3(512), 0, st, gd);
      else if (ask_ks == 2) TG_KLAUNCH((k_gemm<2, 2, 2, 2, true>), dim3((unsigned)grid), dim3(512), 0, st, gd);
      else if (ask_depth == 4) TG_KLAUNCH((k_gemm<2, 2, 1, 4, true>), dim3((unsigned)grid), dim3(256), 0, st, gd);
      else TG_KLAUNCH((k_gemm<2, 2, 1, 2, true>), dim3((unsigned)grid), dim3(256), 0, st, gd);
    }
  } else if (split) {
    no_ride();
    TG_KLAUNCH((k_gemm<2, 2, 2, 2>), dim3((unsigned)grid), dim3(512), 0, st, gd);
  } else if (depth_knob == 2 && rider && live_blocks(g, grid, 64) <= RIDER_MAX_LIVE) {
    WbRider wr = *rider;
    wr.blocks = rider_blocks(live_blocks(g, grid, 64), 256, 2 * wr.a.B);
    wr.last = 1u;
    TG_KLAUNCH((k_gemm_r<WbRider, 2, 2, 1, 2>), dim3((unsigned)grid + wr.blocks), dim3(256), 0, st, gd, wr);
    *rode = true;
  } else if (depth_knob == 2 && collate && live_blocks(g, grid, 64) <= RIDER_MAX_LIVE) {
    CollateRider co = *collate;
    collate_blocks(co);
    TG_KLAUNCH((k_gemm_r<CollateRider, 2, 2, 1, 2>), dim3((unsigned)grid + co.blocks), dim3(256), 0, st, gd, co);
    *rode = true;
  } else if (depth_knob == 2) {
    no_ride();
    TG_KLAUNCH((k_gemm<2, 2, 1, 2>), dim3((unsigned)grid), dim3(256), 0, st, gd);
  } else {
    no_ride();
    TG_KLAUNCH((k_gemm<2, 2, 1, 4>), dim3((unsigned)grid), dim3(256), 0, st, gd);
  }
  return check_launch("gemm");
}

// ---------------------------------------------------------------------------------
// GRU cell, gates fused into the GEMM epilogue.  A block owns 128 rows x 32 hidden
// columns and accumulates four planes per column: r and z over K = [x | h], i_n over x
// only, h_n over h only (no wasted MFMAs on the zero blocks of a packed [4d, 5d] weight).
// ---------------------------------------------------------------------------------
// The last hidden columns of the GRU when d is not a multiple of 32: a 16-column tile on
// v_mfma_f32_16x16x4_f32 instead of a 32-column tile that is mostly padding (d = 172: 12 columns of 32).
// One block owns 144 rows x 16 columns x 3 planes - three quarters of the MFMA work of a 96 x 32 block of
// k_gru<3, 4>, whose launch it shares, but the same 24 KB of operands staged per tile, which is what sets the
// pace (12 wavefronts: 3 row groups of 48 rows x 4 k-groups).
// Lane l feeds A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16] and receives D[4 (l / 16) + r][l % 16].
// Tiles are [row][k] with a 34-float stride: (34 r + k) mod 32 is injective over the 16 rows x 2 k of a
// 32-lane read group.  Plain pipeline (next tile in registers while this one is multiplied).
constexpr int T16_RT = 3;                       // 16-row MFMA tiles per row group
constexpr int T16_RG = 16 * T16_RT;             // rows per row group (three groups)
constexpr int T16_ROWS = 3 * T16_RG, T16_LD = 34;
constexpr int T16_A = 2 * T16_ROWS * T16_LD, T16_B = 2 * 48 * T16_LD;  // floats
__device__ __forceinline__ void gru_tail16(const GruArgs& g, int tb, int j0, float* __restrict__ arena,
                                           float* __restrict__ hs, int* __restrict__ orow_s) {
  constexpr int THREADS = 768;
  float (*As)[T16_ROWS][T16_LD] = reinterpret_cast<float (*)[T16_ROWS][T16_LD]>(arena);
  float (*Bs)[48][T16_LD] = reinterpret_cast<float (*)[48][T16_LD]>(arena + T16_A);
  float (*red)[4][3][16][64] = reinterpret_cast<float (*)[4][3][16][64]>(arena);
  float (*Hs)[17] = reinterpret_cast<float (*)[17]>(hs);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rg = wave % 3, ks = wave / 3;
  const int d = g.d, xw = g.xw;
  int64_t M = g.cap;
  if (g.n_dev) M = min(M, (int64_t)*g.n_dev);
  const int64_t m0 = (int64_t)tb * T16_ROWS;
  if (m0 >= M) return;
  if (tid < T16_ROWS) {
    const int64_t m = min(m0 + tid, M - 1);
    orow_s[tid] = g.out_rows ? g.out_rows[m] : (int)m;
  }
  const int li = lane & 15, lk = lane >> 4;
  const int jb = min(j0 + li, d - 1);
  const float br = g.b_ih[jb] + g.b_hh[jb];
  const float bz = g.b_ih[d + jb] + g.b_hh[d + jb];
  const float bin = g.b_ih[2 * d + jb], bhn = g.b_hh[2 * d + jb];
  // staging: two activation float4 per thread (rows ar, ar + 96), one weight float4 for the first 384 threads
  const int ar = tid >> 3, ac4 = (tid & 7) * 4;
  const float* xrow[2];
  const float* hrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = min(m0 + min(ar + 96 * i, T16_ROWS - 1), M - 1);
    xrow[i] = g.x.p + (g.x.idx ? g.x.idx[m] : m) * g.x.ld;
    hrow[i] = g.h.p + (g.h.idx ? g.h.idx[m] : m) * g.h.ld;
  }
  const int wl = min(ar, 47);  // weight-tile row: plane wl / 16, column wl % 16
  const int wj = min(j0 + (wl & 15), d - 1);
  const float* wx = g.w_ih + ((int64_t)(wl >> 4) * d + wj) * xw;
  const float* wh = g.w_hh + ((int64_t)(wl >> 4) * d + wj) * d;
  const int nkx = (xw + BK - 1) / BK - g.x_skip_n, nkh = (d + BK - 1) / BK;  // message tiles that are processed
  const int nkt = nkx + nkh;
  // processed message tile t holds k-tile t, or t + x_skip_n past the skipped run: a column shift on the ADDRESSES of
  // those tiles (xs_sh), the k arithmetic itself runs on the compacted width xwe
  const int xs_at = g.x_skip_at, xs_sh = g.x_skip_n * BK, xwe = xw - xs_sh;
  struct Stage {
    float4 a0, a1, b;
  };
  auto load_tile = [&](int t, Stage& r) {  // raw loads from clamped addresses (tiles past the end: the last one again)
    t = min(t, nkt - 1);
    const bool hp = t >= nkx;
    const int k = (hp ? t - nkx : t) * BK + ac4;
    const int kc = (k < (hp ? d : xwe) ? k : 0) + ((!hp && t >= xs_at) ? xs_sh : 0);
    r.a0 = ldg4((hp ? hrow[0] : xrow[0]) + kc);
    r.a1 = ldg4((hp ? hrow[1] : xrow[1]) + kc);
    r.b = ldg4((hp ? wh : wx) + kc);
  };
  auto store_tile = [&](int buf, int t, const Stage& r) {
    const bool hp = t >= nkx;
    const bool kin = (hp ? t - nkx : t) * BK + ac4 < (hp ? d : xwe);
    sts4(As[buf][ar], ac4, kin ? r.a0 : zero4());
    if (ar + 96 < T16_ROWS) sts4(As[buf][ar + 96], ac4, kin ? r.a1 : zero4());
    if (ar < 48) sts4(Bs[buf][ar], ac4, kin ? r.b : zero4());
  };
  f32x4m acc_r[T16_RT], acc_z[T16_RT], acc_in[T16_RT], acc_hn[T16_RT];
#pragma unroll
  for (int i = 0; i < T16_RT; ++i) acc_r[i] = acc_z[i] = acc_in[i] = acc_hn[i] = f32x4m{0.f, 0.f, 0.f, 0.f};
  const int ht = nkx + (j0 >> 5), hc = j0 & 31;  // the memory tile / column offset that holds h[., j0 .. j0 + 16)
  // tile t in LDS[buf]: tile t + 2 is requested into `ld`, the k-steps run, tile t + 1 (`stv`, requested a tile
  // ago) moves to LDS[buf ^ 1]; straight-line body, all loads unconditional
  auto step = [&](auto hp_tag, int buf, int t, Stage& ld, const Stage& stv) {
    constexpr bool HP = decltype(hp_tag)::value;
    load_tile(t + 2, ld);
    float a[2][T16_RT], b[2][3];
    auto read = [&](int st, float* av, float* bv) {  // this k-group's k-step st: columns ks * 8 + 4 st + lk
      const int k = ks * 8 + st * 4 + lk;
#pragma unroll
      for (int rt = 0; rt < T16_RT; ++rt) av[rt] = As[buf][rg * T16_RG + rt * 16 + li][k];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bv[pl] = Bs[buf][pl * 16 + li][k];
    };
    read(0, a[0], b[0]);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int rt = 0; rt < T16_RT; ++rt) {
        acc_r[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st][rt], b[st][0], acc_r[rt], 0, 0, 0);
        if (st == 0 && rt == 0) read(1, a[1], b[1]);
        acc_z[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st][rt], b[st][1], acc_z[rt], 0, 0, 0);
        if (HP) acc_hn[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st][rt], b[st][2], acc_hn[rt], 0, 0, 0);
        else acc_in[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st][rt], b[st][2], acc_in[rt], 0, 0, 0);
      }
    }
    if (HP && t == ht) {
      for (int f = tid; f < T16_ROWS * 16; f += THREADS) Hs[f >> 4][f & 15] = As[buf][f >> 4][hc + (f & 15)];
    }
    store_tile(buf ^ 1, t + 1, stv);
    __syncthreads();
  };
  using HP0 = std::integral_constant<bool, false>;
  using HP1 = std::integral_constant<bool, true>;
  Stage sa, sb;
  load_tile(0, sa);
  load_tile(1, sb);
  store_tile(0, 0, sa);
  __syncthreads();
  int t = 0;
  for (; t + 2 <= nkx; t += 2) {
    step(HP0{}, 0, t, sa, sb);
    step(HP0{}, 1, t + 1, sb, sa);
  }
  if (t < nkx) {  // odd number of message tiles: the memory tiles start in LDS[1]
    step(HP0{}, 0, t, sa, sb);
    for (++t; t + 2 <= nkt; t += 2) {
      step(HP1{}, 1, t, sb, sa);
      step(HP1{}, 0, t + 1, sa, sb);
    }
    if (t < nkt) step(HP1{}, 1, t, sb, sa);
  } else {
    for (; t + 2 <= nkt; t += 2) {
      step(HP1{}, 0, t, sa, sb);
      step(HP1{}, 1, t + 1, sb, sa);
    }
    if (t < nkt) step(HP1{}, 0, t, sa, sb);
  }
  // fold the four k-groups (as in k_gru): groups [half, 2 half) write, groups [0, half) add
#pragma unroll
  for (int half = 2; half >= 1; half /= 2) {
    if (ks >= half && ks < 2 * half) {
#pragma unroll
      for (int rt = 0; rt < T16_RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[ks - half][0][rg][rt * 4 + r][lane] = acc_r[rt][r];
          red[ks - half][1][rg][rt * 4 + r][lane] = acc_z[rt][r];
          red[ks - half][2][rg][rt * 4 + r][lane] = acc_in[rt][r];
          red[ks - half][3][rg][rt * 4 + r][lane] = acc_hn[rt][r];
        }
    }
    __syncthreads();
    if (ks < half) {
#pragma unroll
      for (int rt = 0; rt < T16_RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc_r[rt][r] += red[ks][0][rg][rt * 4 + r][lane];
          acc_z[rt][r] += red[ks][1][rg][rt * 4 + r][lane];
          acc_in[rt][r] += red[ks][2][rg][rt * 4 + r][lane];
          acc_hn[rt][r] += red[ks][3][rg][rt * 4 + r][lane];
        }
    }
    if (half > 1) __syncthreads();
  }
  if (ks != 0) return;
  const int j = j0 + li;
  if (j >= d) return;
#pragma unroll
  for (int rt = 0; rt < T16_RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = rg * T16_RG + rt * 16 + 4 * lk + r;
      const int64_t m = m0 + lr;
      if (m >= M) continue;
      const float rgate = fast_sigmoid(acc_r[rt][r] + br);
      const float zgate = fast_sigmoid(acc_z[rt][r] + bz);
      const float hn = acc_hn[rt][r] + bhn;
      const float ng = fast_tanh(acc_in[rt][r] + bin + rgate * hn);
      const float hv = (1.f - zgate) * ng + zgate * Hs[lr][li];
      g.out[(int64_t)orow_s[lr] * g.ldo + j] = hv;
      if (g.out2) g.out2[(g.out2_by_row ? (int64_t)orow_s[lr] : m) * (int64_t)d + j] = g.add2 ? hv + g.add2[(int64_t)orow_s[lr] * d + j] : hv;
      if (g.gates) {
        float* gp = g.gates + m * 4 * (int64_t)d + j;
        gp[0] = rgate; gp[d] = zgate; gp[2 * d] = ng; gp[3 * d] = hn;
      }
    }
}

// diagnostic only (TG_GRU_DBG & 16): per-block s_memtime stamps {entry, loop start, loop end, exit}
__device__ unsigned long long g_gru_trace[2048 * 4];

template <int NW, int KS>
__global__ void __launch_bounds__(64 * NW * KS, (NW == 4 && KS == 1) ? 2 : 1) k_gru(GruArgs g) {
  // NW wavefronts stack 32-row MFMA tiles (BM = 32 NW rows per block); with KS = 2 a second
  // group of NW wavefronts takes the other half of every tile's k-steps into its own
  // accumulators (summed through LDS at the end), which puts two independent instruction
  // streams on every SIMD: a single wave drives the f32 matrix pipe to only ~65 % here.
  constexpr int THREADS = 64 * NW * KS;
  constexpr int BM = 32 * NW;
  constexpr int RP = THREADS / 8;              // tile rows staged per pass (8 threads per 32-float row)
  constexpr int NA = BM / RP;                  // activation float4 per thread per tile
  constexpr int NBL = (96 + RP - 1) / RP;      // weight float4 per thread per tile (3 planes x 32 rows)
  constexpr int NOPS = NA + NBL;
  constexpr int PP = 8 / KS;                   // k-step pairs per wave per tile
  constexpr int SL = (NOPS + PP / 2 - 1) / (PP / 2);  // memory-op slots per k-step pair: 3 in the wide blocks, 4 in the 32-row one
  static_assert(BM % RP == 0 && SL <= 4, "at most four memory-op slots between the six MFMAs of a k-step pair");
  const unsigned long long t_entry = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  // One LDS arena: the operand tiles while the loop runs, the k-group fold afterwards (the tiles are dead then).
  constexpr int A_FLOATS = 2 * BM * LDK, B_FLOATS = 2 * 3 * 32 * LDK;
  // k-group fold: either halving rounds (the upper half of the live groups writes at once), or - where the LDS allows -
  // one reduce-scatter round after which EVERY k-group finishes the rows it owns (see the epilogue)
  constexpr bool SCAT = KS > 1 && NW != 3;
  constexpr int OWN = 16 / KS;  // accumulator registers (row quads) a k-group owns in the scattered epilogue
  constexpr int RED_FLOATS = KS == 1 ? 0 : SCAT ? NW * KS * (KS - 1) * 4 * OWN * 64 : (KS / 2) * 4 * NW * 16 * 64;
  constexpr int ARENA0 = (A_FLOATS + B_FLOATS) > RED_FLOATS ? (A_FLOATS + B_FLOATS) : RED_FLOATS;
  constexpr bool TAIL = NW == 3 && KS == 4;  // this instance also serves the 16-column tail blocks (gru_tail16)
  constexpr int ARENA = TAIL && (T16_A + T16_B) > ARENA0 ? (T16_A + T16_B) : ARENA0;
  __shared__ float arena[ARENA];
  float (*As)[BM][LDK] = reinterpret_cast<float (*)[BM][LDK]>(arena);
  float (*Bs)[3][32][LDK] = reinterpret_cast<float (*)[3][32][LDK]>(arena + A_FLOATS);
  float (*red)[4][NW][16][64] = reinterpret_cast<float (*)[4][NW][16][64]>(arena);  // [k-group slot][plane][row wave]
  // epilogue operands staged while the loop runs (no global load is left for the epilogue, where its
  // latency would be exposed): the old-memory tile h[m, j0..j0+32) is one of the A tiles the loop
  // streams anyway, the output rows are fetched at block start
  // (128-row blocks: a lane captures the old-memory values of ITS accumulator rows in registers when that tile passes -
  // 17 KB of LDS less, which is what lets two k_gru<4, 1> blocks share a CU)
  constexpr bool HREG = NW == 4;
  constexpr int NHOLD = KS == 1 ? 16 : OWN;
  constexpr int HS_FLOATS = HREG ? 4 : TAIL && T16_ROWS * 17 > BM * LDK ? T16_ROWS * 17 : BM * LDK;
  __shared__ float hs_raw[HS_FLOATS];
  float hold_r[HREG ? NHOLD : 1];
  __shared__ int orow_s[TAIL ? T16_ROWS : BM];
  float (*Hs)[LDK] = reinterpret_cast<float (*)[LDK]>(hs_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rw = wave % NW, ks = wave / NW;
  const int d = g.d, xw = g.xw;
  // column tiles of 32 handled here; with a tail (tail_blocks > 0) the last, partial one belongs to gru_tail16,
  // whose blocks come FIRST in the grid so that they start with everybody else
  const int NT = (d + 31) / 32 - (g.tail_blocks > 0 ? 1 : 0);
  int64_t M = g.cap;
  if (g.n_dev) M = min(M, (int64_t)*g.n_dev);
  const int xcd = blockIdx.x & 7;
  int s = blockIdx.x >> 3;
  if (TAIL && g.tail_blocks > 0) {
    // With the tail the launch is sized to fit the chip in ONE round (every CU at most one block), so the live
    // blocks must also be dealt evenly over the eight XCDs (blockIdx % 8): XCD x works through the row tiles
    // mt = x (mod 8), all their column tiles, then its share of the tail blocks - the XCDs that own one row
    // tile less take NT tail blocks each first, the rest is dealt round robin.  All from the live row count.
    const int MT = (int)((M + BM - 1) / BM), TT = (int)((M + T16_ROWS - 1) / T16_ROWS);
    const int r = MT & 7, n_main = (MT / 8 + (xcd < r ? 1 : 0)) * NT;
    if (s >= n_main) {
      const int ti = s - n_main, light = r ? 8 - r : 0, first = light * NT;
      int tb;
      if (r && xcd >= r) tb = ti < NT ? (xcd - r) * NT + ti : first + xcd + 8 * (ti - NT);
      else tb = first + xcd + 8 * ti;
      if (tb < TT) gru_tail16(g, tb, NT * 32, arena, hs_raw, orow_s);
      return;
    }
  }
  const int64_t mt = (int64_t)(s / NT) * 8 + xcd;
  const int nt = s % NT;
  const int64_t m0 = mt * BM;
  if (m0 >= M) return;
  const int j0 = nt * 32;
  if (tid < BM) {
    const int64_t m = min(m0 + tid, M - 1);
    orow_s[tid] = g.out_rows ? g.out_rows[m] : (int)m;
  }
  // gate biases, requested before the loop
  const int jb = min(j0 + (lane & 31), d - 1);
  const float br = g.b_ih[jb] + g.b_hh[jb];
  const float bz = g.b_ih[d + jb] + g.b_hh[d + jb];
  const float bin = g.b_ih[2 * d + jb], bhn = g.b_hh[2 * d + jb];
  const int ar = tid >> 3, ac4 = (tid & 7) * 4;
  // branch-free staging (see k_gemm): clamped addresses, zeros only for k past the segment
  const float* xrow[NA];
  const float* hrow[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int64_t m = min(m0 + ar + i * RP, M - 1);
    xrow[i] = g.x.p + (g.x.idx ? g.x.idx[m] : m) * g.x.ld;
    hrow[i] = g.h.p + (g.h.idx ? g.h.idx[m] : m) * g.h.ld;
  }
  const int nkx = (xw + BK - 1) / BK - g.x_skip_n, nkh = (d + BK - 1) / BK;  // message tiles that are processed
  const int nkt = nkx + nkh;
  // processed message tile t holds k-tile t, or t + x_skip_n past the skipped run: a column shift on the ADDRESSES of
  // those tiles (xs_sh), the k arithmetic itself runs on the compacted width xwe
  const int xs_at = g.x_skip_at, xs_sh = g.x_skip_n * BK, xwe = xw - xs_sh;
  float4 ra0[NA], rb0[NBL], ra1[NA], rb1[NBL];
  const int fr = lane & 31, fk = lane >> 5;
  f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc_r[i] = acc_z[i] = acc_in[i] = acc_hn[i] = 0.f;
  const int dbg = g.dbg;  // bit 16: record s_memtime stamps (diagnostic knob, 0 in production)
  // ---- hand-scheduled tile step ------------------------------------------------------
  // A CU pulls only ~10 B/clk through its vector-memory path, and a wave issues in order: a
  // burst of global loads (or ds_writes) in front of the MFMAs stalls the matrix pipe until
  // the memory queue drains (measured: loop = MFMA + loads + stores, nothing hidden).  So the
  // memory work of the OTHER tiles is threaded between the MFMAs of this tile and the order
  // is pinned with sched_barrier: during the first half of a wave's k-step pairs the global
  // loads of tile t+2, during the second half the ds_writes of tile t+1 into the other LDS
  // buffer, while the operand fragments of the next pair are read one pair ahead.
  struct Frag {
    float a0, a1, b00, b01, b10, b11, b20, b21;
  };
  auto read_pair = [&](int buf, int p, Frag& f) {  // k-steps 2p and 2p+1 of the tile in LDS[buf]
    const float* ap = &As[buf][rw * 32 + fr][fk + 4 * p];
    const float* b0 = &Bs[buf][0][fr][fk + 4 * p];
    const float* b1 = &Bs[buf][1][fr][fk + 4 * p];
    const float* b2 = &Bs[buf][2][fr][fk + 4 * p];
    f.a0 = ap[0]; f.a1 = ap[2];
    f.b00 = b0[0]; f.b01 = b0[2];
    f.b10 = b1[0]; f.b11 = b1[2];
    f.b20 = b2[0]; f.b21 = b2[2];
  };
  auto load_one = [&](int t, int i, float4* ra, float4* rb) {  // i-th staged float4 of tile t
    const bool hp = t >= nkx;
    const int k = (hp ? t - nkx : t) * BK + ac4;
    const int width = hp ? d : xw;  // row stride of the weight operand
    const int kc = (k < (hp ? d : xwe) ? k : 0) + ((!hp && t >= xs_at) ? xs_sh : 0);
    // raw load from a clamped address; columns past the segment are zeroed when the tile is
    // written to LDS (store_one), so nothing consumes the load result here
    if (i < NA) {
      ra[i] = ldg4((hp ? hrow[i] : xrow[i]) + kc);
    } else {
      const int L = min(ar + (i - NA) * RP, 95);  // row of the [3 planes x 32] weight tile
      const int jc = min(j0 + (L & 31), d - 1);
      rb[i - NA] = ldg4((hp ? g.w_hh : g.w_ih) + ((int64_t)(L >> 5) * d + jc) * width + kc);
    }
  };
  auto store_one = [&](int buf, int t, int i, const float4* ra, const float4* rb) {  // tile t's i-th float4
    const bool hp = t >= nkx;
    const bool kin = (hp ? t - nkx : t) * BK + ac4 < (hp ? d : xwe);
    if (i < NA) {
      sts4(As[buf][ar + i * RP], ac4, kin ? ra[i] : zero4());
    } else {
      const int L = ar + (i - NA) * RP;
      if (L < 96) sts4(Bs[buf][L >> 5][L & 31], ac4, kin ? rb[i - NA] : zero4());
    }
  };
#define TG_SB() __builtin_amdgcn_sched_barrier(0)
  auto tile = [&](auto hp_tag, int buf, int t, float4* la, float4* lb, const float4* sa, const float4* sb) {
    constexpr bool HP = decltype(hp_tag)::value;
    const int tl = min(t + 2, nkt - 1);  // past the end: a redundant reload keeps the block branch-free
    const int p0 = ks * PP;              // this wave's share of the tile's k-step pairs
    Frag cur, nxt;
    read_pair(buf, p0, cur);
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      const int op0 = (q % (PP / 2)) * SL;  // SL memory-op slots per pair; NOPS of them are used
      auto memop = [&](int i) {
        if (i < NOPS) {
          if (q < PP / 2) load_one(tl, i, la, lb);
          else store_one(buf ^ 1, t + 1, i, sa, sb);
        }
      };
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0, cur.b00, acc_r, 0, 0, 0);
      if (q < PP - 1) read_pair(buf, p0 + q + 1, nxt);
      TG_SB();
      acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0, cur.b10, acc_z, 0, 0, 0);
      memop(op0);
      TG_SB();
      if (HP) acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0, cur.b20, acc_hn, 0, 0, 0);
      else acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0, cur.b20, acc_in, 0, 0, 0);
      if (SL > 3) memop(op0 + 3);
      TG_SB();
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1, cur.b01, acc_r, 0, 0, 0);
      memop(op0 + 1);
      TG_SB();
      acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1, cur.b11, acc_z, 0, 0, 0);
      TG_SB();
      if (HP) acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1, cur.b21, acc_hn, 0, 0, 0);
      else acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1, cur.b21, acc_in, 0, 0, 0);
      memop(op0 + 2);
      TG_SB();
      cur = nxt;
    }
    if (HP && t == nkx + nt) {  // this A tile is h[m0.., j0..j0+32): keep it for the epilogue
      if constexpr (HREG) {
#pragma unroll
        for (int q = 0; q < NHOLD; ++q) {
          const int r = KS == 1 ? q : ks * OWN + q;
          hold_r[q] = As[buf][rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk][fr];
        }
      } else {
        for (int f = tid; f < BM * 32; f += THREADS) Hs[f >> 5][f & 31] = As[buf][f >> 5][f & 31];
      }
    }
    __syncthreads();
  };
#undef TG_SB
  using HP0 = std::integral_constant<bool, false>;
  using HP1 = std::integral_constant<bool, true>;
  // prologue: tile 0 -> LDS[0]; tile 1 -> registers R1
#pragma unroll
  for (int i = 0; i < NOPS; ++i) load_one(0, i, ra0, rb0);
#pragma unroll
  for (int i = 0; i < NOPS; ++i) load_one(min(1, nkt - 1), i, ra1, rb1);
#pragma unroll
  for (int i = 0; i < NOPS; ++i) store_one(0, 0, i, ra0, rb0);
  __syncthreads();
  const unsigned long long t_loop0 = (dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  // Message tiles (i_n plane) first, then memory tiles (h_n plane), each as its own straight-line loop
  // over tile pairs, with the memory loops written out for both LDS-buffer parities.  A single loop with a
  // per-tile phase test moved the accumulators between registers on every path (64 v_mov_b64 per pair)
  // and its joins made the compiler wait for prefetched tiles half a tile early.
  // even tile: multiply LDS[0]; load tile t+2 -> R0; store tile t+1 (R1) -> LDS[1]; odd tile: mirrored
  int t = 0;
  for (; t + 2 <= nkx; t += 2) {
    tile(HP0{}, 0, t, ra0, rb0, ra1, rb1);
    tile(HP0{}, 1, t + 1, ra1, rb1, ra0, rb0);
  }
  if (t < nkx) {  // odd number of message tiles: the memory tiles start in LDS[1]
    tile(HP0{}, 0, t, ra0, rb0, ra1, rb1);
    for (++t; t + 2 <= nkt; t += 2) {
      tile(HP1{}, 1, t, ra1, rb1, ra0, rb0);
      tile(HP1{}, 0, t + 1, ra0, rb0, ra1, rb1);
    }
    if (t < nkt) tile(HP1{}, 1, t, ra1, rb1, ra0, rb0);
  } else {
    for (; t + 2 <= nkt; t += 2) {
      tile(HP1{}, 0, t, ra0, rb0, ra1, rb1);
      tile(HP1{}, 1, t + 1, ra1, rb1, ra0, rb0);
    }
    if (t < nkt) tile(HP1{}, 0, t, ra0, rb0, ra1, rb1);
  }
  const unsigned long long t_loop1 = (dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  const int j = min(j0 + fr, d - 1);
  const bool jok = j0 + fr < d;
  auto finish = [&](int r, int hq, float ar_, float az_, float ain_, float ahn_) {  // gates + blend of accumulator row r (hq: its slot in hold_r)
    const int lr = rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
    const int64_t m = m0 + lr;
    const float hold = HREG ? hold_r[HREG ? hq : 0] : Hs[lr][fr];
    const int64_t orow = orow_s[lr];
    const float rg = fast_sigmoid(ar_ + br);
    const float zg = fast_sigmoid(az_ + bz);
    const float hn = ahn_ + bhn;
    const float ng = fast_tanh(ain_ + bin + rg * hn);
    if (jok && m < M) {
      const float hv = (1.f - zg) * ng + zg * hold;
      g.out[orow * g.ldo + j] = hv;
      if (g.out2) g.out2[(g.out2_by_row ? orow : m) * (int64_t)d + j] = g.add2 ? hv + g.add2[orow * d + j] : hv;
      if (g.gates) {
        float* gp = g.gates + m * 4 * (int64_t)d + j;
        gp[0] = rg; gp[d] = zg; gp[2 * d] = ng; gp[3 * d] = hn;
      }
    }
  };
  if constexpr (SCAT) {
    // Reduce-scatter over the k-groups of a row wave: group v owns accumulator registers [v OWN, (v+1) OWN) (a quarter
    // of the tile's rows with four groups).  Every group parks the registers the others own in LDS - one round, one
    // barrier - then sums its own share and runs the gate arithmetic and the stores for it, so the epilogue's
    // transcendental work is spread over all the block's wavefronts instead of the first k-group's.
    float (*sc)[KS][KS - 1][4][OWN][64] = reinterpret_cast<float (*)[KS][KS - 1][4][OWN][64]>(arena);
#pragma unroll
    for (int v = 0; v < KS; ++v) {
      if (v != ks) {  // wave-uniform
        const int slot = (ks - v - 1 + KS) % KS;
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
          sc[rw][v][slot][0][q][lane] = acc_r[v * OWN + q];
          sc[rw][v][slot][1][q][lane] = acc_z[v * OWN + q];
          sc[rw][v][slot][2][q][lane] = acc_in[v * OWN + q];
          sc[rw][v][slot][3][q][lane] = acc_hn[v * OWN + q];
        }
      }
    }
    __syncthreads();
    // The partial sums of a row are added in k-group order 0, 1, .., KS - 1 WHICHEVER group owns the row: a row's result
    // must not depend on its place in the tile (the eager updater's row order comes from an atomic compaction and
    // differs from run to run; summed owner-first the step was reproducible only to the last bit or two).
    float o_r[OWN], o_z[OWN], o_in[OWN], o_hn[OWN];
#pragma unroll
    for (int gsrc = 0; gsrc < KS; ++gsrc) {
      float t_r[OWN], t_z[OWN], t_in[OWN], t_hn[OWN];
      if (gsrc == ks) {  // wave-uniform
#pragma unroll
        for (int v = 0; v < KS; ++v)
          if (v == ks) {
#pragma unroll
            for (int q = 0; q < OWN; ++q) {
              t_r[q] = acc_r[v * OWN + q]; t_z[q] = acc_z[v * OWN + q];
              t_in[q] = acc_in[v * OWN + q]; t_hn[q] = acc_hn[v * OWN + q];
            }
          }
      } else {
        const int sl = (gsrc - ks - 1 + KS) % KS;
#pragma unroll
        for (int q = 0; q < OWN; ++q) {
          t_r[q] = sc[rw][ks][sl][0][q][lane]; t_z[q] = sc[rw][ks][sl][1][q][lane];
          t_in[q] = sc[rw][ks][sl][2][q][lane]; t_hn[q] = sc[rw][ks][sl][3][q][lane];
        }
      }
#pragma unroll
      for (int q = 0; q < OWN; ++q) {
        o_r[q] = gsrc == 0 ? t_r[q] : o_r[q] + t_r[q];
        o_z[q] = gsrc == 0 ? t_z[q] : o_z[q] + t_z[q];
        o_in[q] = gsrc == 0 ? t_in[q] : o_in[q] + t_in[q];
        o_hn[q] = gsrc == 0 ? t_hn[q] : o_hn[q] + t_hn[q];
      }
    }
#pragma unroll
    for (int q = 0; q < OWN; ++q) finish(ks * OWN + q, q, o_r[q], o_z[q], o_in[q], o_hn[q]);
  } else {
  // fold the k-groups' partial sums into group 0, halving the number of live groups per round: groups
  // [half, 2 half) write, groups [0, half) add (the last tile's barrier has retired every read of the tiles)
#pragma unroll
  for (int half = KS / 2; half >= 1; half /= 2) {
    if (ks >= half && ks < 2 * half) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[ks - half][0][rw][r][lane] = acc_r[r];
        red[ks - half][1][rw][r][lane] = acc_z[r];
        red[ks - half][2][rw][r][lane] = acc_in[r];
        red[ks - half][3][rw][r][lane] = acc_hn[r];
      }
    }
    __syncthreads();
    if (ks < half) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_r[r] += red[ks][0][rw][r][lane];
        acc_z[r] += red[ks][1][rw][r][lane];
        acc_in[r] += red[ks][2][rw][r][lane];
        acc_hn[r] += red[ks][3][rw][r][lane];
      }
    }
    if (half > 1) __syncthreads();  // the next round overwrites the slots
  }
  if (ks == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) finish(r, r, acc_r[r], acc_z[r], acc_in[r], acc_hn[r]);
  }
  }
  if ((dbg & 16) && tid == 0 && blockIdx.x < 2048) {
    g_gru_trace[blockIdx.x * 4 + 0] = t_entry;
    g_gru_trace[blockIdx.x * 4 + 1] = t_loop0;
    g_gru_trace[blockIdx.x * 4 + 2] = t_loop1;
    g_gru_trace[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  }
}

// ---------------------------------------------------------------------------------
// The GRU cell for FEW rows (the eager updater of a C2-sized batch: ~1 000 rows): 32 rows x 32 hidden columns
// per block, four wavefronts, and NO LDS in the k-loop.
// In a 32-row block the four wavefronts share nothing: with one row wave, LDS staging only re-shapes global
// rows into MFMA fragments, and that costs more than the MFMAs (s_memtime ablations of k_gru<1, 4>: 44.9 k
// cycles per block loop, 27.2 k without the LDS stores, 25.1 k for the MFMAs alone - the store path moves
// 64-79 B/clk per CU and every ds_write holds the issuing wave).  The sum over k does not care which k values
// share an MFMA step, so a lane can feed the matrix unit straight from what it loads: lane (row r, half kh)
// reads the 64 contiguous bytes A[r][k0 + 16 kh .. + 15] of its row as four float4, the same slice of its weight
// row in each of the three planes, and MFMA step (q, j) multiplies element j of float4 q - k = k0 + 16 kh + 4 q + j
// on both operands.  The k-tiles are dealt to the wavefronts (wave w takes tiles w, w + 4, ...), each into its own
// accumulators: no barrier and no LDS until the fold, a tile is 16 loads and 48 MFMAs of one wavefront, the next
// tile's loads are in flight meanwhile.  Fold and epilogue as k_gru's scattered form: every wavefront finishes
// a quarter of the rows; the old-memory values and output rows of those are requested at kernel start.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gru_direct(GruArgs g) {
  constexpr int KS = 4, OWN = 4;
  __shared__ float sc_raw[KS * (KS - 1) * 4 * OWN * 64];
  float (*sc)[KS - 1][4][OWN][64] = reinterpret_cast<float (*)[KS - 1][4][OWN][64]>(sc_raw);
  const unsigned long long t_entry = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
  const int fr = lane & 31, fk = lane >> 5;
  const int d = g.d, xw = g.xw;
  const int NT = (d + 31) / 32;
  int64_t M = g.cap;
  if (g.n_dev) M = min(M, (int64_t)*g.n_dev);
  const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
  const int64_t mt = (int64_t)(s / NT) * 8 + xcd;
  const int nt = s % NT;
  const int64_t m0 = mt * 32;
  if (m0 >= M) return;
  const int j0 = nt * 32;
  const int jc = min(j0 + fr, d - 1);
  const bool jok = j0 + fr < d;
  // this lane's operand rows: activation row m0 + fr, weight row jc of every plane
  const int64_t mrow = min(m0 + fr, M - 1);
  const float* xrow = g.x.p + (g.x.idx ? g.x.idx[mrow] : mrow) * g.x.ld;
  const float* hrow = g.h.p + (g.h.idx ? g.h.idx[mrow] : mrow) * g.h.ld;
  const float* wi = g.w_ih + (int64_t)jc * xw;
  const float* wh = g.w_hh + (int64_t)jc * d;
  const int64_t wi_ps = (int64_t)d * xw, wh_ps = (int64_t)d * d;  // plane strides
  // epilogue operands of the rows this wavefront finishes (accumulator registers [4 ks, 4 ks + 4))
  float hold[OWN], addv[OWN];  // (addv: GruArgs.add2 of the row's node, requested with the old memory value - same depth)
  int64_t orow[OWN];
#pragma unroll
  for (int q = 0; q < OWN; ++q) {
    const int64_t mm = min(m0 + 8 * ks + q + 4 * fk, M - 1);
    const int64_t node = g.h.idx ? g.h.idx[mm] : mm;
    hold[q] = g.h.p[node * g.h.ld + jc];
    addv[q] = (g.out2 && g.add2) ? g.add2[node * d + jc] : 0.f;
    orow[q] = g.out_rows ? (int64_t)g.out_rows[mm] : mm;
  }
  const float br = g.b_ih[jc] + g.b_hh[jc];
  const float bz = g.b_ih[d + jc] + g.b_hh[d + jc];
  const float bin = g.b_ih[2 * d + jc], bhn = g.b_hh[2 * d + jc];
  const int nkx = (xw + BK - 1) / BK - g.x_skip_n, nkh = (d + BK - 1) / BK;
  const int nkt = nkx + nkh;
  const int xs_at = g.x_skip_at, xs_sh = g.x_skip_n * BK, xwe = xw - xs_sh;  // zero k-tiles skipped (see k_gru)
  // this wavefront's tiles: message tiles ks, ks + 4, ... < nkx, then memory tiles th0, th0 + 4, ... < nkt
  const int nx = ks < nkx ? (nkx - ks + 3) / 4 : 0;
  const int th0 = nkx + ((ks - nkx) % 4 + 4) % 4;
  const int nh = th0 < nkt ? (nkt - th0 + 3) / 4 : 0;
  const int n_my = nx + nh;
  auto tile_of = [&](int i) { return i < nx ? ks + 4 * i : th0 + 4 * (i - nx); };
  struct Tile {
    float4 a[4], w0[4], w1[4], w2[4];
  };
  // raw loads from clamped addresses; `live` bit q: float4 q of the activation slice lies inside the operand (the
  // others are taken as zero when they are used - the weight slice then multiplies zeros, whatever it holds)
  auto load_tile = [&](int i, Tile& T, unsigned& live) {
    // past the end: a redundant reload keeps the code branch-free; a wavefront without tiles (fewer than four k-tiles
    // in all) reads tile th0 >= nkt, whose every k is out of range: clamped addresses, nothing live
    const int t = tile_of(max(0, min(i, n_my - 1)));
    const bool hp = t >= nkx;
    const int kb = (hp ? t - nkx : t) * BK + 16 * fk;
    const int sh = (!hp && t >= xs_at) ? xs_sh : 0;
    const int wid = hp ? d : xwe;
    const float* ar = hp ? hrow : xrow;
    const float* wr = hp ? wh : wi;
    const int64_t ps = hp ? wh_ps : wi_ps;
    live = 0u;
    int kc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kb + 4 * q;
      if (k < wid) live |= 1u << q;
      kc[q] = (k < wid ? k : 0) + sh;
    }
    // the four float4 of one row slice are requested back to back (one cache line each row)
#pragma unroll
    for (int q = 0; q < 4; ++q) T.a[q] = ldg4(ar + kc[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) T.w0[q] = ldg4(wr + kc[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) T.w1[q] = ldg4(wr + ps + kc[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) T.w2[q] = ldg4(wr + 2 * ps + kc[q]);
  };
  f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc_r[i] = acc_z[i] = acc_in[i] = acc_hn[i] = 0.f;
  auto mma_tile = [&](auto hp_tag, const Tile& T, unsigned live) {
    constexpr bool HP = decltype(hp_tag)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = ((live >> q) & 1u) ? T.a[q] : zero4();
      const float av[4] = {a.x, a.y, a.z, a.w};
      const float b0[4] = {T.w0[q].x, T.w0[q].y, T.w0[q].z, T.w0[q].w};
      const float b1[4] = {T.w1[q].x, T.w1[q].y, T.w1[q].z, T.w1[q].w};
      const float b2[4] = {T.w2[q].x, T.w2[q].y, T.w2[q].z, T.w2[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b0[j], acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b1[j], acc_z, 0, 0, 0);
        if (HP) acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b2[j], acc_hn, 0, 0, 0);
        else acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b2[j], acc_in, 0, 0, 0);
      }
    }
  };
  using HP0 = std::integral_constant<bool, false>;
  using HP1 = std::integral_constant<bool, true>;
  Tile T0, T1;
  unsigned l0 = 0u, l1 = 0u;
  load_tile(0, T0, l0);
  const unsigned long long t_loop0 = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  // One step: request tile i + 1 into the idle register set, THEN multiply tile i (the order is pinned: left to
  // itself the scheduler sinks the loads to the end of the MFMA stream, where the next tile waits for them in full).
  // Message tiles (i_n plane) first, then memory tiles (h_n plane), each as straight-line pairs, written out for both
  // register parities of the phase change.
#define TG_STEP(HPT, CUR, LCUR, NXT, LNXT, INEXT)     \
  do {                                                \
    load_tile(INEXT, NXT, LNXT);                      \
    __builtin_amdgcn_sched_barrier(0);                \
    mma_tile(HPT{}, CUR, LCUR);                       \
    __builtin_amdgcn_sched_barrier(0);                \
  } while (0)
  int i = 0;
  for (; i + 2 <= nx; i += 2) {
    TG_STEP(HP0, T0, l0, T1, l1, i + 1);
    TG_STEP(HP0, T1, l1, T0, l0, i + 2);
  }
  if (i < nx) {  // odd number of message tiles: the memory tiles start in the other register set
    TG_STEP(HP0, T0, l0, T1, l1, i + 1);
    for (++i; i + 2 <= n_my; i += 2) {
      TG_STEP(HP1, T1, l1, T0, l0, i + 1);
      TG_STEP(HP1, T0, l0, T1, l1, i + 2);
    }
    if (i < n_my) mma_tile(HP1{}, T1, l1);
  } else {
    for (; i + 2 <= n_my; i += 2) {
      TG_STEP(HP1, T0, l0, T1, l1, i + 1);
      TG_STEP(HP1, T1, l1, T0, l0, i + 2);
    }
    if (i < n_my) mma_tile(HP1{}, T0, l0);
  }
#undef TG_STEP
  const unsigned long long t_loop1 = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  // reduce-scatter over the four wavefronts (as k_gru's scattered epilogue): park what the others own, sum one's own
#pragma unroll
  for (int v = 0; v < KS; ++v) {
    if (v != ks) {
      const int slot = (ks - v - 1 + KS) % KS;
#pragma unroll
      for (int q = 0; q < OWN; ++q) {
        sc[v][slot][0][q][lane] = acc_r[v * OWN + q];
        sc[v][slot][1][q][lane] = acc_z[v * OWN + q];
        sc[v][slot][2][q][lane] = acc_in[v * OWN + q];
        sc[v][slot][3][q][lane] = acc_hn[v * OWN + q];
      }
    }
  }
  __syncthreads();
  // in wavefront order 0..3 whichever wavefront owns the row (see k_gru: the result must not depend on the row's place)
  float o_r[OWN], o_z[OWN], o_in[OWN], o_hn[OWN];
#pragma unroll
  for (int gsrc = 0; gsrc < KS; ++gsrc) {
    float t_r[OWN], t_z[OWN], t_in[OWN], t_hn[OWN];
    if (gsrc == ks) {  // wave-uniform
#pragma unroll
      for (int v = 0; v < KS; ++v)
        if (v == ks) {
#pragma unroll
          for (int q = 0; q < OWN; ++q) {
            t_r[q] = acc_r[v * OWN + q]; t_z[q] = acc_z[v * OWN + q];
            t_in[q] = acc_in[v * OWN + q]; t_hn[q] = acc_hn[v * OWN + q];
          }
        }
    } else {
      const int sl = (gsrc - ks - 1 + KS) % KS;
#pragma unroll
      for (int q = 0; q < OWN; ++q) {
        t_r[q] = sc[ks][sl][0][q][lane]; t_z[q] = sc[ks][sl][1][q][lane];
        t_in[q] = sc[ks][sl][2][q][lane]; t_hn[q] = sc[ks][sl][3][q][lane];
      }
    }
#pragma unroll
    for (int q = 0; q < OWN; ++q) {
      o_r[q] = gsrc == 0 ? t_r[q] : o_r[q] + t_r[q];
      o_z[q] = gsrc == 0 ? t_z[q] : o_z[q] + t_z[q];
      o_in[q] = gsrc == 0 ? t_in[q] : o_in[q] + t_in[q];
      o_hn[q] = gsrc == 0 ? t_hn[q] : o_hn[q] + t_hn[q];
    }
  }
#pragma unroll
  for (int q = 0; q < OWN; ++q) {
    const int64_t m = m0 + 8 * ks + q + 4 * fk;
    const float rg = fast_sigmoid(o_r[q] + br);
    const float zg = fast_sigmoid(o_z[q] + bz);
    const float hn = o_hn[q] + bhn;
    const float ng = fast_tanh(o_in[q] + bin + rg * hn);
    if (jok && m < M) {
      const float hv = (1.f - zg) * ng + zg * hold[q];
      g.out[orow[q] * g.ldo + j0 + fr] = hv;
      if (g.out2) g.out2[(g.out2_by_row ? orow[q] : m) * (int64_t)d + j0 + fr] = hv + addv[q];
      if (g.gates) {
        float* gp = g.gates + m * 4 * (int64_t)d + j0 + fr;
        gp[0] = rg; gp[d] = zg; gp[2 * d] = ng; gp[3 * d] = hn;
      }
    }
  }
  if ((g.dbg & 16) && tid == 0 && blockIdx.x < 2048) {
    g_gru_trace[blockIdx.x * 4 + 0] = t_entry;
    g_gru_trace[blockIdx.x * 4 + 1] = t_loop0;
    g_gru_trace[blockIdx.x * 4 + 2] = t_loop1;
    g_gru_trace[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  }
}

// ---- LDS-free updater blocks on 16 x 16 MFMA tiles: 16 RT rows x 16 hidden columns ------------------------------------
// k_gru_direct's 32 x 32 blocks leave CUs idle whenever (row tiles x column tiles) is not close to 256: C2's ~1 060 rows at
// d = 172 are 34 x 6 = 204 blocks, and the launch lasts as long as ONE block.  On v_mfma_f32_16x16x4_f32 (same flops per
// cycle) the hidden width is cut into 16-column tiles (172 -> 11 tiles, 2 % padding instead of 10 %) and a block owns 16,
// 32 or 48 rows: the kernel picks, from the LIVE row count, the smallest of the three whose blocks all fit the chip at
// once (least work per block; C2: 23 x 11 = 253 blocks of 48 rows, three quarters of the work of a 32 x 32 block each;
// the batch-of-200 workload: 32-row blocks).  Same scheme otherwise: no LDS in the k-loop, a lane feeds the matrix unit
// from what it loads (lane (row i, quarter kq) reads the 32 contiguous bytes A[i][k0 + 8 kq ..] of each of its RT rows
// and of its weight row in each plane; MFMA step (q, j) multiplies element j of float4 q on both operands - the sum over
// k does not care which k values share a step), the k-tiles are dealt to the four wavefronts, whose accumulators meet
// in a reduce-scatter through LDS (k-group order: bit-reproducible).  Blocks are dealt to the XCDs in contiguous chunks of
// the (row tile, column tile) sequence - balanced to within one block, and a row tile's gathered rows are fetched by one
// XCD's L2, two at a chunk border.
template <int RT>
__device__ __forceinline__ void gru_direct16_body(const GruArgs& g, int64_t M, int64_t mt, int nt, float* sc_raw) {
  constexpr int KS = 4;
  TG_PT(const unsigned long long pt_entry = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;)  // (diagnostic, as in gemm_ks16_tile)
  float (*sc)[KS - 1][4][RT][64] = reinterpret_cast<float (*)[KS - 1][4][RT][64]>(sc_raw);  // [owner][slot][plane][row tile]
  const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int d = g.d, xw = g.xw;
  const int64_t m0 = mt * (16 * RT);
  const int j0 = nt * 16;
  const int jc = min(j0 + li, d - 1);
  const bool jok = j0 + li < d;
  const float* xrow[RT];
  const float* hrow[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t mrow = min(m0 + 16 * rt + li, M - 1);
    xrow[rt] = g.x.p + (g.x.idx ? g.x.idx[mrow] : mrow) * g.x.ld;
    hrow[rt] = g.h.p + (g.h.idx ? g.h.idx[mrow] : mrow) * g.h.ld;
  }
  const float* wi = g.w_ih + (int64_t)jc * xw;
  const float* wh = g.w_hh + (int64_t)jc * d;
  const int64_t wi_ps = (int64_t)d * xw, wh_ps = (int64_t)d * d;  // plane strides
  // epilogue operands of the rows this wavefront finishes: accumulator register ks of every row tile (row 4 lk + ks)
  float hold[RT], addv[RT];
  int64_t orow[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t mm = min(m0 + 16 * rt + 4 * lk + ks, M - 1);
    const int64_t node = g.h.idx ? g.h.idx[mm] : mm;
    hold[rt] = g.h.p[node * g.h.ld + jc];
    addv[rt] = (g.out2 && g.add2) ? g.add2[node * d + jc] : 0.f;
    orow[rt] = g.out_rows ? (int64_t)g.out_rows[mm] : mm;
  }
  const float br = g.b_ih[jc] + g.b_hh[jc];
  const float bz = g.b_ih[d + jc] + g.b_hh[d + jc];
  const float bin = g.b_ih[2 * d + jc], bhn = g.b_hh[2 * d + jc];
  const int nkx = (xw + BK - 1) / BK - g.x_skip_n, nkh = (d + BK - 1) / BK;
  const int nkt = nkx + nkh;
  const int xs_at = g.x_skip_at, xs_sh = g.x_skip_n * BK, xwe = xw - xs_sh;  // zero k-tiles skipped (see k_gru)
  // this wavefront's tiles: message tiles ks, ks + 4, ... < nkx, then memory tiles th0, th0 + 4, ... < nkt
  const int nx = ks < nkx ? (nkx - ks + 3) / 4 : 0;
  const int th0 = nkx + ((ks - nkx) % 4 + 4) % 4;
  const int nh = th0 < nkt ? (nkt - th0 + 3) / 4 : 0;
  const int n_my = nx + nh;
  auto tile_of = [&](int i) { return i < nx ? ks + 4 * i : th0 + 4 * (i - nx); };
  struct Tile {
    float4 a[RT][2], w0[2], w1[2], w2[2];
  };
  auto load_tile = [&](int i, Tile& T, unsigned& live) {  // raw loads from clamped addresses (see k_gru_direct)
    const int t = tile_of(max(0, min(i, n_my - 1)));
    const bool hp = t >= nkx;
    const int kb = (hp ? t - nkx : t) * BK + 8 * lk;
    const int sh = (!hp && t >= xs_at) ? xs_sh : 0;
    const int wid = hp ? d : xwe;
    const float* wr = hp ? wh : wi;
    const int64_t ps = hp ? wh_ps : wi_ps;
    live = 0u;
    int kc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = kb + 4 * q;
      if (k < wid) live |= 1u << q;
      kc[q] = (k < wid ? k : 0) + sh;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 2; ++q) T.a[rt][q] = ldg4((hp ? hrow[rt] : xrow[rt]) + kc[q]);
#pragma unroll
    for (int q = 0; q < 2; ++q) T.w0[q] = ldg4(wr + kc[q]);
#pragma unroll
    for (int q = 0; q < 2; ++q) T.w1[q] = ldg4(wr + ps + kc[q]);
#pragma unroll
    for (int q = 0; q < 2; ++q) T.w2[q] = ldg4(wr + 2 * ps + kc[q]);
  };
  f32x4m acc_r[RT], acc_z[RT], acc_in[RT], acc_hn[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc_r[rt] = acc_z[rt] = acc_in[rt] = acc_hn[rt] = f32x4m{0.f, 0.f, 0.f, 0.f};
  auto mma_tile = [&](auto hp_tag, const Tile& T, unsigned live) {
    constexpr bool HP = decltype(hp_tag)::value;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float av[RT][4];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float4 a = ((live >> q) & 1u) ? T.a[rt][q] : zero4();
        av[rt][0] = a.x; av[rt][1] = a.y; av[rt][2] = a.z; av[rt][3] = a.w;
      }
      const float b0[4] = {T.w0[q].x, T.w0[q].y, T.w0[q].z, T.w0[q].w};
      const float b1[4] = {T.w1[q].x, T.w1[q].y, T.w1[q].z, T.w1[q].w};
      const float b2[4] = {T.w2[q].x, T.w2[q].y, T.w2[q].z, T.w2[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc_r[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][j], b0[j], acc_r[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc_z[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][j], b1[j], acc_z[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if (HP) acc_hn[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][j], b2[j], acc_hn[rt], 0, 0, 0);
          else acc_in[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][j], b2[j], acc_in[rt], 0, 0, 0);
        }
      }
    }
  };
  using HP0 = std::integral_constant<bool, false>;
  using HP1 = std::integral_constant<bool, true>;
  Tile T0, T1;
  unsigned l0 = 0u, l1 = 0u;
  load_tile(0, T0, l0);
#define TG_STEP(HPT, CUR, LCUR, NXT, LNXT, INEXT)     \
  do {                                                \
    load_tile(INEXT, NXT, LNXT);                      \
    __builtin_amdgcn_sched_barrier(0);                \
    mma_tile(HPT{}, CUR, LCUR);                       \
    __builtin_amdgcn_sched_barrier(0);                \
  } while (0)
  TG_PT(const unsigned long long pt_loop0 = (g.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull;)
  int i = 0;
  for (; i + 2 <= nx; i += 2) {
    TG_STEP(HP0, T0, l0, T1, l1, i + 1);
    TG_STEP(HP0, T1, l1, T0, l0, i + 2);
  }
  if (i < nx) {  // odd number of message tiles: the memory tiles start in the other register set
    TG_STEP(HP0, T0, l0, T1, l1, i + 1);
    for (++i; i + 2 <= n_my; i += 2) {
      TG_STEP(HP1, T1, l1, T0, l0, i + 1);
      TG_STEP(HP1, T0, l0, T1, l1, i + 2);
    }
    if (i < n_my) mma_tile(HP1{}, T1, l1);
  } else {
    for (; i + 2 <= n_my; i += 2) {
      TG_STEP(HP1, T0, l0, T1, l1, i + 1);
      TG_STEP(HP1, T1, l1, T0, l0, i + 2);
    }
    if (i < n_my) mma_tile(HP1{}, T0, l0);
  }
#undef TG_STEP
  TG_PT(unsigned long long pt_loop1 = 0ull; if (g.dbg & 16) {
    asm volatile("s_nop 0" ::"v"(acc_r[0][0]));
    pt_loop1 = __builtin_amdgcn_s_memtime();
  })
  // reduce-scatter over the four wavefronts: wavefront v finishes accumulator register v of every row tile; the others'
  // registers are parked in LDS (one round, one barrier)
#pragma unroll
  for (int v = 0; v < KS; ++v) {
    if (v != ks) {  // wave-uniform
      const int slot = (ks - v - 1 + KS) % KS;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        sc[v][slot][0][rt][lane] = acc_r[rt][v];
        sc[v][slot][1][rt][lane] = acc_z[rt][v];
        sc[v][slot][2][rt][lane] = acc_in[rt][v];
        sc[v][slot][3][rt][lane] = acc_hn[rt][v];
      }
    }
  }
  __syncthreads();
  // partial sums are added in wavefront order 0..3 whichever wavefront owns the row (k_gru: a row's result must not
  // depend on its place in the tile)
  float o_r[RT], o_z[RT], o_in[RT], o_hn[RT];
#pragma unroll
  for (int gsrc = 0; gsrc < KS; ++gsrc) {
    float t_r[RT], t_z[RT], t_in[RT], t_hn[RT];
    if (gsrc == ks) {  // wave-uniform
#pragma unroll
      for (int v = 0; v < KS; ++v)
        if (v == ks) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            t_r[rt] = acc_r[rt][v]; t_z[rt] = acc_z[rt][v]; t_in[rt] = acc_in[rt][v]; t_hn[rt] = acc_hn[rt][v];
          }
        }
    } else {
      const int sl = (gsrc - ks - 1 + KS) % KS;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        t_r[rt] = sc[ks][sl][0][rt][lane]; t_z[rt] = sc[ks][sl][1][rt][lane];
        t_in[rt] = sc[ks][sl][2][rt][lane]; t_hn[rt] = sc[ks][sl][3][rt][lane];
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      o_r[rt] = gsrc == 0 ? t_r[rt] : o_r[rt] + t_r[rt];
      o_z[rt] = gsrc == 0 ? t_z[rt] : o_z[rt] + t_z[rt];
      o_in[rt] = gsrc == 0 ? t_in[rt] : o_in[rt] + t_in[rt];
      o_hn[rt] = gsrc == 0 ? t_hn[rt] : o_hn[rt] + t_hn[rt];
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t m = m0 + 16 * rt + 4 * lk + ks;
    const float rg = fast_sigmoid(o_r[rt] + br);
    const float zg = fast_sigmoid(o_z[rt] + bz);
    const float hn = o_hn[rt] + bhn;
    const float ng = fast_tanh(o_in[rt] + bin + rg * hn);
    if (jok && m < M) {
      const float hv = (1.f - zg) * ng + zg * hold[rt];
      g.out[orow[rt] * g.ldo + j0 + li] = hv;
      if (g.out2) g.out2[(g.out2_by_row ? orow[rt] : m) * (int64_t)d + j0 + li] = hv + addv[rt];
      if (g.gates) {
        float* gp = g.gates + m * 4 * (int64_t)d + j0 + li;
        gp[0] = rg; gp[d] = zg; gp[2 * d] = ng; gp[3 * d] = hn;
      }
    }
  }
  TG_PT(if ((g.dbg & 16) && tid == 0 && blockIdx.x < 2048) {
    g_gru_trace[blockIdx.x * 4 + 0] = pt_entry;
    g_gru_trace[blockIdx.x * 4 + 1] = pt_loop0;
    g_gru_trace[blockIdx.x * 4 + 2] = pt_loop1;
    g_gru_trace[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  })
}

constexpr int GRU16_RT_MAX = 6;
// RTM: the most rows / 16 this instance offers (3: 16 / 32 / 48 rows, 176 + 72 registers; 6: also 64 / 96, 254 + 144 - the
// bigger register file costs the 48-row blocks 5 %, so launches whose bound fits 48-row blocks take the small instance)
template <int RTM>
__global__ void __launch_bounds__(256) k_gru_direct16(GruArgs g) {
  __shared__ float sc_raw[4 * 3 * 4 * RTM * 64];
  int64_t M = g.cap;
  if (g.n_dev) M = min(M, (int64_t)*g.n_dev);
  if (M <= 0) return;
  const int NT = (g.d + 15) / 16;
  // rows per block from the LIVE row count: the smallest of 16 / 32 / 48 (/ 64 / 96) whose blocks fit the 256 CUs at once
  int rt = RTM;
  if (((M + 15) / 16) * NT <= 256) rt = 1;
  else if (((M + 31) / 32) * NT <= 256) rt = 2;
  else if (RTM > 3 && ((M + 47) / 48) * NT <= 256) rt = 3;
  else if (RTM > 3 && ((M + 63) / 64) * NT <= 256) rt = 4;
  const int64_t total = ((M + 16 * rt - 1) / (16 * rt)) * NT;
  // XCD x (blockIdx % 8) works through the chunk [x per, (x + 1) per) of the tile sequence.  The grid is 256 blocks whatever
  // the capacity (no tail of dead blocks: the row CAPACITY of the eager updater is twice its live rows and more); a row
  // count beyond the largest blocks' single round makes blocks take a second tile
  const int64_t per = (total + 7) / 8;
  for (int64_t jx = blockIdx.x >> 3; jx < per; jx += gridDim.x >> 3) {
    const int64_t b = (int64_t)(blockIdx.x & 7) * per + jx;
    if (b >= total) break;
    const int64_t mt = b / NT;
    const int nt = (int)(b - mt * NT);
    if (rt == 1) gru_direct16_body<1>(g, M, mt, nt, sc_raw);
    else if (rt == 2) gru_direct16_body<2>(g, M, mt, nt, sc_raw);
    else if (RTM == 3 || rt == 3) gru_direct16_body<3>(g, M, mt, nt, sc_raw);
    else if (rt == 4) gru_direct16_body<(RTM > 3 ? 4 : 3)>(g, M, mt, nt, sc_raw);
    else gru_direct16_body<RTM>(g, M, mt, nt, sc_raw);
    __syncthreads();  // the fold's LDS is re-used by the next tile
  }
}

extern "C" int tg_debug_gru_trace(unsigned long long* out_host, int n_blocks) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_gru_trace), sizeof(unsigned long long) * 4 * n_blocks) == hipSuccess ? 0 : -4;
}

int gru_launch(const GruArgs& g, hipStream_t st) {
  if (g.cap <= 0) return TG_OK;
  if (g.d <= 0 || (g.d % 4) || g.xw <= 0 || (g.xw % 4)) return TG_EINVAL;
  static const int dbg = getenv("TG_GRU_DBG") ? atoi(getenv("TG_GRU_DBG")) : 0;  // bit 16: trace stamps
  static const int force_nw = getenv("TG_GRU_NW") ? atoi(getenv("TG_GRU_NW")) : 0;  // tuning knob
  GruArgs a = g;
  a.dbg = dbg;
  const int NT = (g.d + 31) / 32;
  // small problems (at most ~64k live rows): 64-row blocks double the block count so that two
  // blocks share a CU and cover each other's stalls; large ones keep 128 rows (half the weight traffic)
  static const int ks_knob = getenv("TG_GRU_KS") ? atoi(getenv("TG_GRU_KS")) : 0;  // tuning knob: 0 = by the grid (below)
  // 96-row blocks (four k-groups of three row waves: 12 wavefronts, three per SIMD) with the partial column tile
  // as 16-column blocks: a quarter less work per block and no padded columns, which pays exactly when the whole
  // launch fits the chip in ONE round - every CU runs at most one block, so the duration is one block's duration
  // (C2 shapes, 4174 rows: 61.9 -> 51.8 us).  One block more than CUs and it costs a second round (4400 rows:
  // 83 us), so the choice needs a guaranteed bound on the live rows: rows_hint (the caller's bound, e.g. the node
  // count), not the capacity.  At C2's steady state (4350-4550 involved nodes of 9228) it does not apply.
  // Smaller still when it fits: 64-row blocks (k_gru<2, 4>, 8 wavefronts; 1588 rows at d = 172: 36 us against
  // 47 us with 96 rows and 58 us with 128).  Its blocks follow the plain XCD map, so every XCD must hold its
  // share: ceil(row tiles / 8) x column tiles <= 32 CUs.
  bool small = false, tiny = false, micro = false;
  {
    const int tail = g.d % 32;
    const bool use_tail = tail > 0 && tail <= 16;
    const int64_t rows = g.rows_hint > 0 ? std::min<int64_t>(g.rows_hint, g.cap) : g.cap;
    const int64_t blocks96 = cdiv(rows, 96) * (NT - (use_tail ? 1 : 0)) + (use_tail ? cdiv(rows, T16_ROWS) : 0);
    small = blocks96 + 8 <= 256;  // (+8: the per-XCD dealing can leave one XCD a block short of full)
    tiny = cdiv(cdiv(rows, 64), 8) * NT <= 32;
    // fewer rows still: 32-row blocks (k_gru<1, 4>, four wavefronts) double the block count once more - the updater of the
    // eager step runs on ~1000 rows at C2: 17 x 6 = 102 blocks of 64 rows leave 154 CUs idle, 34 x 6 = 204 do not.  An XCD
    // that ends up with a few more blocks than its 32 CUs co-hosts two of these small blocks on a CU (no second round)
    micro = cdiv(cdiv(rows, 32), 8) * NT <= 40;
  }
  static const int micro_knob = getenv("TG_GRU_MICRO") ? atoi(getenv("TG_GRU_MICRO")) : 1;  // tuning knob: 0 = off
  {
    // LDS-free blocks of 16 hidden columns x 16 .. 96 rows (k_gru_direct16 picks the rows per block from the live row
    // count): whenever the caller's bound on the rows says that the 96-row blocks fit the chip at once.  Measured against
    // the kernels below (d = 172): 380 rows 18.5 -> 11.3 us, 1 060 rows 19.9 -> 16.1 us; d = 100, 1 060 rows 14.7 -> 9.3 us
    static const int d16_knob = getenv("TG_GRU_D16") ? atoi(getenv("TG_GRU_D16")) : 1;  // tuning knob: 0 = off, 3 = always
    const int64_t rows_b = g.rows_hint > 0 ? std::min<int64_t>(g.rows_hint, g.cap) : g.cap;
    const int NT16 = (g.d + 15) / 16;
    if (force_nw == 0 && d16_knob && (d16_knob == 3 || cdiv(rows_b, 16 * GRU16_RT_MAX) * NT16 <= 256)) {
      a.tail_blocks = 0;
      // (the sampler riders of the collate prefetch were tried here too - the sampler reads the graph only - and cost the
      // updater more than they saved the query-row launch: C2 updater +4.6 us, launch behind it -0.7 us; C4 +22 us)
      const dim3 grid16((unsigned)std::min<int64_t>(256, 8 * cdiv(cdiv(g.cap, 16) * NT16, 8)));
      if (cdiv(rows_b, 48) * NT16 <= 256) TG_KLAUNCH(k_gru_direct16<3>, grid16, dim3(256), 0, st, a);
      else TG_KLAUNCH(k_gru_direct16<GRU16_RT_MAX>, grid16, dim3(256), 0, st, a);
      return check_launch("gru(16 x 16)");
    }
  }
  if (force_nw == 1 || (force_nw == 0 && micro && micro_knob)) {
    a.tail_blocks = 0;
    static const int direct_knob = getenv("TG_GRU_DIRECT") ? atoi(getenv("TG_GRU_DIRECT")) : 1;  // tuning knob: 0 = LDS-staged
    const dim3 grid32((unsigned)(8 * cdiv(cdiv(g.cap, 32), 8) * NT));
    if (direct_knob) TG_KLAUNCH(k_gru_direct, grid32, dim3(256), 0, st, a);
    else TG_KLAUNCH((k_gru<1, 4>), grid32, dim3(256), 0, st, a);
    return check_launch("gru(32)");
  }
  if (force_nw == 2 || (force_nw == 0 && tiny)) {
    a.tail_blocks = 0;
    TG_KLAUNCH((k_gru<2, 4>), dim3((unsigned)(8 * cdiv(cdiv(g.cap, 64), 8) * NT)), dim3(512), 0, st, a);
    return check_launch("gru(64)");
  }
  if (force_nw == 3 || (force_nw == 0 && small)) {
    const int tail = g.d % 32;
    const bool use_tail = tail > 0 && tail <= 16;  // the partial column tile as 16-column blocks of T16_ROWS rows
    a.tail_blocks = use_tail ? (int)cdiv(g.cap, T16_ROWS) : 0;
    const int ntm = NT - (use_tail ? 1 : 0);
    // per XCD: its row tiles x column tiles, then at most NT + ceil(tails / 8) + 1 tail slots (see the kernel's map)
    const int64_t per_xcd = cdiv(cdiv(g.cap, 96), 8) * ntm + (use_tail ? ntm + cdiv(a.tail_blocks, 8) + 1 : 0);
    TG_KLAUNCH((k_gru<3, 4>), dim3((unsigned)(8 * per_xcd)), dim3(768), 0, st, a);
    return check_launch("gru(96)");
  }
  // 128-row blocks.  More blocks than CUs: four wavefronts per block and TWO blocks per CU (k_gru<4, 1>: 58 KB of LDS - the
  // old-memory tile of the epilogue lives in registers - and at most 256 registers), so that one block's prologue (first
  // tiles exposed) and epilogue (gates, scattered stores) run under the other's k-loop, and there is no k-group fold.
  // Measured against the eight-wavefront blocks (k_gru<4, 2>, one per CU), rows x message width -> d:
  // 65 536 x 1 024 -> 256 1 200 -> 1 073 us (120 TF/s); 49 152 x 688 -> 172 485 -> 435 us; 8 192 x 1 024 -> 256 152 -> 140 us;
  // a launch of at most one block per CU keeps the eight wavefronts (4 096 x 1 024 -> 256: 77 against 81 us).
  const int64_t grid = 8 * cdiv(cdiv(g.cap, 128), 8) * NT;
  const int64_t live = g.rows_hint > 0 ? cdiv(std::min<int64_t>(g.rows_hint, g.cap), 128) * NT : grid;
  // (the caller's row bound carries a margin - 1.5 x the rows seen: between one and 1.25 blocks per CU by that bound the launch
  // usually fits one round, where the eight wavefronts are ahead: 5 000 x 688 -> 172 61.6 against 64.2 us)
  if (ks_knob == 2 || (ks_knob == 0 && live <= (g.rows_hint > 0 ? 320 : 256)))
    TG_KLAUNCH((k_gru<4, 2>), dim3((unsigned)grid), dim3(512), 0, st, a);
  else
    TG_KLAUNCH((k_gru<4, 1>), dim3((unsigned)grid), dim3(256), 0, st, a);
  return check_launch("gru");
}

// ---------------------------------------------------------------------------------
// Weight gradients: out[n, k] = sum_m Y[m, n] X[m, k].  Both MFMA operands are read along m,
// so tiles are staged [m][col] exactly as they lie in memory (no transposition): lane l feeds
// A[n = l&31][m = l>>5] = Ys[m][n], B[m = l>>5][k = l&31] = Xs[m][k].
// ---------------------------------------------------------------------------------
constexpr int TN_T = 64;        // output tile (n and k extent)
constexpr int TN_MC = 32;       // m rows per staged chunk
constexpr int TN_LD = TN_T + 32;  // row stride: the two half-waves (rows m, m+1) hit disjoint banks

__device__ __forceinline__ void tn_block(const TnArgs& a, int splits, int b) {
  __shared__ float Ys[2][TN_MC][TN_LD];
  __shared__ float Xs[2][TN_MC][TN_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = (a.n + TN_T - 1) / TN_T, KT = (a.k + TN_T - 1) / TN_T;
  const int kt = b % KT; b /= KT;
  const int nt = b % NT; b /= NT;
  const int bz = b % a.nbatch;
  const int sp = b / a.nbatch;
  int64_t M = a.m_cap;
  if (a.m_dev) M = min(M, (int64_t)*a.m_dev);
  const int64_t mc = ((M + splits - 1) / splits + TN_MC - 1) / TN_MC * TN_MC;  // rows per split
  const int64_t m_lo = (int64_t)sp * mc, m_hi = min(M, m_lo + mc);
  const int n0 = nt * TN_T, k0 = kt * TN_T;
  const int wn = wave >> 1, wk = wave & 1;
  const int fr = lane & 31, fk = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // staging coordinates: 16 threads per 64-float row, 16 rows per pass, 2 passes per operand
  const int sr = tid >> 4, sc = (tid & 15) * 4;
  const float* yb = a.y + (int64_t)bz * a.y_bs;
  const float* x0b = a.x0.p + (int64_t)bz * a.x0_bs;
  const int ycol = min(n0 + sc, a.n - 4);           // clamped: columns past N are zeroed at the LDS store
  const bool yin = n0 + sc < a.n;
  const int kcol = k0 + sc;
  const bool xin = kcol < a.k;
  const int kc = xin ? kcol : 0;
  const bool seg1 = kc >= a.x0.w;
  // Two chunks travel in registers (sets A / B) while a third is multiplied from LDS, and the row indices of
  // gathered X rows (mailbox / memory rows of the outdated nodes) are fetched one chunk further ahead, so that
  // a chunk's row loads never wait for their index.  All loads are unconditional (rows clamped to M - 1, zeroed
  // at the LDS store when past the split): the loop body is straight-line code with exact wait counts.
  struct Regs {
    float4 y[2], x[2];
    int64_t row[2];
  };
  const int64_t* xidx = seg1 ? a.x1.idx : a.x0.idx;
  auto load_index = [&](int64_t mb, Regs& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = min(mb + sr + i * 16, M - 1);
      r.row[i] = xidx ? xidx[m] : m;
    }
  };
  auto load_chunk = [&](int64_t mb, Regs& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = min(mb + sr + i * 16, M - 1);
      r.y[i] = ldg4(yb + m * a.ldy + ycol);
      const float* xr = seg1 ? a.x1.p + r.row[i] * a.x1.ld + (kc - a.x0.w) : x0b + r.row[i] * a.x0.ld + kc;
      r.x[i] = ldg4(xr);
    }
  };
  auto store_chunk = [&](int buf, int64_t mb, const Regs& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool live = mb + sr + i * 16 < m_hi;
      *reinterpret_cast<float4*>(&Ys[buf][sr + i * 16][sc]) = (live && yin) ? r.y[i] : zero4();
      *reinterpret_cast<float4*>(&Xs[buf][sr + i * 16][sc]) = (live && xin) ? r.x[i] : zero4();
    }
  };
  const bool do_bias = a.bias_out && kt == 0 && tid < TN_T;  // column sums of Y ride on the k-tile-0 blocks
  float bsum = 0.f;
  // chunk `mb` is in LDS[buf]: request chunk mb + 2 into `ld` (its indices are there already) and the indices of
  // chunk mb + 3 into `nx`'s slot... (see the call sites), multiply, then move chunk mb + 1 from `stv` to LDS
  auto step = [&](int buf, int64_t mb, Regs& ld, const Regs& stv) {
    load_chunk(mb + 2 * TN_MC, ld);
    const float* yp = &Ys[buf][fk][wn * 32 + fr];
    const float* xp = &Xs[buf][fk][wk * 32 + fr];
#pragma unroll
    for (int s2 = 0; s2 < TN_MC / 2; ++s2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yp[s2 * 2 * TN_LD], xp[s2 * 2 * TN_LD], acc, 0, 0, 0);
    if (do_bias) {
      if (a.bias_rs) {
        for (int r = 0; r < TN_MC; ++r) {
          const int64_t m = min(mb + r, M - 1);  // rows past m_hi are zero in Ys
          bsum = fmaf(Ys[buf][r][tid], a.bias_rs[m * a.ld_brs + a.brs_col + bz], bsum);
        }
      } else {
#pragma unroll
        for (int r = 0; r < TN_MC; ++r) bsum += Ys[buf][r][tid];
      }
    }
    store_chunk(buf ^ 1, mb + TN_MC, stv);
    __syncthreads();
  };
  if (m_lo < m_hi) {
    Regs ra, rb;
    load_index(m_lo, ra);
    load_index(m_lo + TN_MC, rb);
    load_chunk(m_lo, ra);
    load_chunk(m_lo + TN_MC, rb);
    store_chunk(0, m_lo, ra);
    load_index(m_lo + 2 * TN_MC, ra);
    __syncthreads();
    int64_t mb = m_lo;
    const int64_t m_pairs = m_lo + (m_hi - m_lo + TN_MC - 1) / TN_MC / 2 * (2 * TN_MC);  // end of the whole chunk pairs
    for (; mb < m_pairs; mb += 2 * TN_MC) {
      // even chunk from LDS[0]: chunk mb+2 -> ra, chunk mb+1 (rb) -> LDS[1]; then rb's indices for chunk mb+3
      step(0, mb, ra, rb);
      load_index(mb + 3 * TN_MC, rb);
      step(1, mb + TN_MC, rb, ra);
      load_index(mb + 4 * TN_MC, ra);
    }
    if (mb < m_hi) step(0, mb, ra, rb);
  }
  // partial tile -> part[sp][bz][n][k] (zeros when this split is empty)
  float* pp = a.part + ((int64_t)sp * a.nbatch + bz) * a.n * a.k;
  const int kk = k0 + wk * 32 + fr;
  if (kk < a.k) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
      if (n < a.n) pp[(int64_t)n * a.k + kk] = acc[r];
    }
  }
  if (do_bias && n0 + tid < a.n) {
    float* bp = a.part + (int64_t)splits * a.nbatch * a.n * a.k;  // bias partials follow the weight partials
    bp[((int64_t)sp * a.nbatch + bz) * a.n + n0 + tid] = bsum;
  }
}

__global__ void __launch_bounds__(256) k_gemm_tn(TnArgs a, int splits) { tn_block(a, splits, (int)blockIdx.x); }

// fixed-order reduction of the split partials of one problem; threads [start, start + stride, ...)
__device__ __forceinline__ void tn_reduce_part(const TnArgs& a, int splits, int64_t start, int64_t stride) {
  const int64_t per = (int64_t)a.n * a.k, total = per * a.nbatch;
  for (int64_t t = start; t < total; t += stride) {
    const int bz = (int)(t / per);
    const int64_t e = t - (int64_t)bz * per;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += a.part[((int64_t)sp * a.nbatch + bz) * per + e];
    float* o = a.out + (int64_t)bz * a.out_bs + (e / a.k) * a.ldo + (e % a.k);
    *o = a.alpha * s + (a.accumulate ? *o : 0.f);
  }
  if (a.bias_out) {
    const float* bp = a.part + (int64_t)splits * a.nbatch * per;
    const int64_t tb = (int64_t)a.n * a.nbatch;
    for (int64_t t = start; t < tb; t += stride) {
      const int bz = (int)(t / a.n);
      const int n = (int)(t - (int64_t)bz * a.n);
      float s = 0.f;
      for (int sp = 0; sp < splits; ++sp) s += bp[((int64_t)sp * a.nbatch + bz) * a.n + n];
      float* o = a.bias_out + (int64_t)bz * a.bias_bs + n;
      *o = a.alpha * s + (a.bias_accumulate ? *o : 0.f);
    }
  }
}
__global__ void k_tn_reduce(TnArgs a, int splits) {
  tn_reduce_part(a, splits, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// several weight-gradient products in ONE launch (and one reduction launch): the nine products of
// the contrastive backward pass are each too small to fill the chip and independent of one another
__global__ void __launch_bounds__(256) k_gemm_tn_group(TnGroup g) {
  int p = 0;
  while (p + 1 < g.n && (int)blockIdx.x >= g.first_block[p + 1]) ++p;
  tn_block(g.a[p], g.splits[p], (int)blockIdx.x - g.first_block[p]);
}
__global__ void k_tn_reduce_group(TnGroup g) {
  int p = 0;
  while (p + 1 < g.n && (int)blockIdx.x >= g.first_rblock[p + 1]) ++p;
  const int nb = g.first_rblock[p + 1] - g.first_rblock[p];
  tn_reduce_part(g.a[p], g.splits[p], (int64_t)((int)blockIdx.x - g.first_rblock[p]) * blockDim.x + threadIdx.x,
                 (int64_t)nb * blockDim.x);
}

int gemm_tn_launch(const TnArgs& a, hipStream_t st) {
  if (a.m_cap <= 0) return TG_OK;
  if (a.n <= 0 || a.k <= 0 || (a.n % 4) || (a.k % 4) || (a.x0.w % 4) || (a.ldy % 4) || a.nbatch <= 0) return TG_EINVAL;
  if (a.x0.w + (a.x1.p ? a.x1.w : 0) != a.k) return TG_EINVAL;
  const int NT = (int)cdiv(a.n, TN_T), KT = (int)cdiv(a.k, TN_T);
  const int64_t tiles = (int64_t)NT * KT * a.nbatch;
  // enough blocks to fill the chip, but a short fixed-order reduction (k_tn_reduce walks the splits serially)
  int64_t splits = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(cdiv(768, tiles), 16), cdiv(a.m_cap, 2 * TN_MC)));
  const int64_t fit = (int64_t)(a.part_floats / ((size_t)a.nbatch * a.n * (a.k + 1)));
  if (fit < 1) return TG_EWORKSPACE;
  splits = std::min(splits, fit);
  TG_KLAUNCH(k_gemm_tn, dim3((unsigned)(tiles * splits)), dim3(256), 0, st, a, (int)splits);
  TG_KLAUNCH(k_tn_reduce, dim3(flat_grid((int64_t)a.n * a.k * a.nbatch, 256)), dim3(256), 0, st, a, (int)splits);
  return check_launch("gemm_tn");
}

int gemm_tn_group_launch(const TnArgs* list, int n, float* part, size_t part_floats, hipStream_t st) {
  if (n <= 0) return TG_OK;
  if (n > TN_GROUP_MAX) return TG_EINVAL;
  TnGroup g{};
  g.n = n;
  int64_t tiles_total = 0;
  for (int p = 0; p < n; ++p) {
    const TnArgs& a = list[p];
    if (a.m_cap <= 0 || a.n <= 0 || a.k <= 0 || (a.n % 4) || (a.k % 4) || (a.x0.w % 4) || (a.ldy % 4) || a.nbatch <= 0)
      return TG_EINVAL;
    if (a.x0.w + (a.x1.p ? a.x1.w : 0) != a.k) return TG_EINVAL;
    tiles_total += cdiv(a.n, TN_T) * cdiv(a.k, TN_T) * a.nbatch;
  }
  // splits: aim at ~1500 blocks in total, at least two staged chunks per block
  const int64_t want = std::max<int64_t>(1, std::min<int64_t>(16, cdiv(1536, tiles_total)));
  size_t off = 0;
  for (int p = 0; p < n; ++p) {
    g.a[p] = list[p];
    TnArgs& a = g.a[p];
    const int64_t tiles = cdiv(a.n, TN_T) * cdiv(a.k, TN_T) * a.nbatch;
    const int64_t sp = std::max<int64_t>(1, std::min<int64_t>(want, cdiv(a.m_cap, 2 * TN_MC)));
    const size_t need = (size_t)sp * a.nbatch * a.n * (a.k + 1);
    if (off + need > part_floats) return TG_EWORKSPACE;
    a.part = part + off;
    a.part_floats = need;
    off += (need + 3) & ~(size_t)3;
    g.splits[p] = (int)sp;
    g.first_block[p + 1] = g.first_block[p] + (int)(tiles * sp);
    g.first_rblock[p + 1] = g.first_rblock[p] + (int)std::min<int64_t>(cdiv((int64_t)a.n * a.k * a.nbatch, 256), 256);
  }
  TG_KLAUNCH(k_gemm_tn_group, dim3((unsigned)g.first_block[n]), dim3(256), 0, st, g);
  TG_KLAUNCH(k_tn_reduce_group, dim3((unsigned)g.first_rblock[n]), dim3(256), 0, st, g);
  return check_launch("gemm_tn_group");
}

// column sums: one block per (64 columns, split of m); partials then a fixed-order reduce
__global__ void __launch_bounds__(256) k_colsum(int64_t m_cap, const int32_t* __restrict__ m_dev, int n,
                                                const float* __restrict__ y, int64_t ldy, float* __restrict__ part,
                                                int splits) {
  __shared__ float red[4][64];
  const int c = blockIdx.x % ((n + 63) / 64) * 64 + (threadIdx.x & 63);
  const int sp = blockIdx.x / ((n + 63) / 64);
  int64_t M = m_cap;
  if (m_dev) M = min(M, (int64_t)*m_dev);
  const int64_t mc = (M + splits - 1) / splits;
  const int64_t lo = sp * mc, hi = min(M, lo + mc);
  float s = 0.f;
  if (c < n)
    for (int64_t m = lo + (threadIdx.x >> 6); m < hi; m += 4) s += y[m * ldy + c];
  red[threadIdx.x >> 6][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && c < n) part[(int64_t)sp * n + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void k_colsum_reduce(int n, const float* __restrict__ part, int splits, float alpha, float* __restrict__ out,
                                int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float s = 0.f;
  for (int sp = 0; sp < splits; ++sp) s += part[(int64_t)sp * n + c];
  out[c] = alpha * s + (accumulate ? out[c] : 0.f);
}

int colsum_launch(int64_t m_cap, const int32_t* m_dev, int n, const float* y, int64_t ldy, float alpha, float* out,
                  int accumulate, float* part, size_t part_floats, hipStream_t st) {
  if (m_cap <= 0 || n <= 0) return TG_OK;
  const int ct = (n + 63) / 64;
  int64_t splits = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(cdiv(512, ct), 16), cdiv(m_cap, 64)));
  splits = std::min<int64_t>(splits, (int64_t)(part_floats / (size_t)n));
  if (splits < 1) return TG_EWORKSPACE;
  TG_KLAUNCH(k_colsum, dim3((unsigned)(ct * splits)), dim3(256), 0, st, m_cap, m_dev, n, y, ldy, part, (int)splits);
  TG_KLAUNCH(k_colsum_reduce, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, n, part, (int)splits, alpha, out,
                     accumulate);
  return check_launch("colsum");
}

}  // namespace tg
