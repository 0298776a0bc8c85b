// EXPERIMENT, not part of libtiger_hip.so: the GRU cell with the activation operand fetched from global memory
// straight into the MFMA operand registers (no LDS for the 128 x 32 activation tile: half of the LDS writes of
// k_gru), weights staged through LDS as in k_gru.  Dropped into tg_gemm.hip next to k_gru (same GruArgs and
// block -> tile map, selected by an environment knob) it passed the parity tests (GRU operator, C2 full size,
// 16-batch soak, fused stream step).  Measured on MI355X (rocprofv3): C2 59.8 us against 60.4 us for k_gru<4,2>;
// C5-scaled run 7.29 M against 7.26 M events/s.  Halving the LDS writes buys nothing: what costs ~10 of the 60 us
// is bringing ~28 KB per tile into the CU at all, not the instruction that carries it (see also
// gru_direct_to_lds.hip and the loop ablation in DESIGN.md section 5).
// ---------------------------------------------------------------------------------
// GRU cell, activations straight into the MFMA operand registers.  The ablation of k_gru shows the LDS write
// port as the cost of staging (~30 KB per tile and CU).  The activation rows (16 of those 30 KB) do not
// need LDS at all: an MFMA lane wants A[row = lane % 32][k] for ITS wave's 32 rows only, so each lane
// fetches 16-byte chunks of its own row from global memory directly (the rows are gathered mailbox /
// memory rows anyway).  To use whole chunks the k order inside every group of eight columns is permuted:
// quad j, k-step e takes column 8 j + e from the half-wave fk = 0 and column 8 j + 4 + e from fk = 1 - for
// both operands alike.  Only the weight tile goes through LDS (12 KB per tile, padded rows, conflict-free
// ds_read2_b32), staged as in k_gru.  Two k-groups of four row waves as in k_gru<4, 2>.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_gru_ra(GruArgs g) {
  constexpr int NW = 4, THREADS = 512, BM = 128;
  constexpr int RP = THREADS / 8;  // weight-tile rows staged per pass (8 threads per 32-float row)
  constexpr int NBL = 2;           // weight float4 per thread per tile (96 rows: the second pass is half empty)
  __shared__ float Bs[2][3][32][LDK];
  __shared__ float red[4][NW][16][64];
  __shared__ float Hs[BM][LDK];
  __shared__ int orow_s[BM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rw = wave % NW, ks = wave / NW;
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
  if (tid < BM) {
    const int64_t m = min(m0 + tid, M - 1);
    orow_s[tid] = g.out_rows ? g.out_rows[m] : (int)m;
  }
  const int fr = lane & 31, fk = lane >> 5;
  const int jb = min(j0 + fr, d - 1);
  const float br = g.b_ih[jb] + g.b_hh[jb];
  const float bz = g.b_ih[d + jb] + g.b_hh[d + jb];
  const float bin = g.b_ih[2 * d + jb], bhn = g.b_hh[2 * d + jb];
  // this lane's activation row (message part, memory part)
  const int arow = rw * 32 + fr;
  const int64_t am = min(m0 + arow, M - 1);
  const float* xrow = g.x.p + (g.x.idx ? g.x.idx[am] : am) * g.x.ld;
  const float* hrow = g.h.p + (g.h.idx ? g.h.idx[am] : am) * g.h.ld;
  const int acol = (ks * 4 + fk) * 4;  // first of this lane's two chunks in a tile: quads 2 ks, 2 ks + 1 -> +0, +8
  const int ar = tid >> 3, ac4 = (tid & 7) * 4;  // weight-tile staging coordinates
  const int nkx = (xw + BK - 1) / BK, nkh = (d + BK - 1) / BK;
  const int nkt = nkx + nkh;
  struct ARegs {
    float4 q[2];
  };
  auto load_a = [&](int t, ARegs& r) {  // this lane's two chunks of tile t (zeros past the segment)
    const bool hp = t >= nkx;
    const int kb = (hp ? t - nkx : t) * BK + acol;
    const int width = hp ? d : xw;
    const float* row = hp ? hrow : xrow;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int k = kb + 8 * qq;
      r.q[qq] = ldg4(row + (k < width ? k : 0));  // raw: chunks past the segment are zeroed where they are used
    }
  };
  auto load_b = [&](int t, int i, float4* rb) {
    const bool hp = t >= nkx;
    const int k = (hp ? t - nkx : t) * BK + ac4;
    const int width = hp ? d : xw;
    const int kc = k < width ? k : 0;
    const int L = min(ar + i * RP, 95);  // row of the [3 planes x 32] weight tile
    const int jc = min(j0 + (L & 31), d - 1);
    rb[i] = ldg4((hp ? g.w_hh : g.w_ih) + ((int64_t)(L >> 5) * d + jc) * width + kc);
  };
  auto store_b = [&](int buf, int t, int i, const float4* rb) {
    const bool hp = t >= nkx;
    const bool kin = (hp ? t - nkx : t) * BK + ac4 < (hp ? d : xw);
    const int L = ar + i * RP;
    if (L < 96) sts4(Bs[buf][L >> 5][L & 31], ac4, kin ? rb[i] : zero4());
  };
  f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc_r[i] = acc_z[i] = acc_in[i] = acc_hn[i] = 0.f;
  struct BFrag {
    float b0[4], b1[4], b2[4];
  };
  auto read_b = [&](int buf, int qq, BFrag& f) {  // columns 8 j + 4 fk .. + 3 of the weight tile, j = 2 ks + qq
    const int k0 = 8 * (2 * ks + qq) + 4 * fk;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f.b0[e] = Bs[buf][0][fr][k0 + e];
      f.b1[e] = Bs[buf][1][fr][k0 + e];
      f.b2[e] = Bs[buf][2][fr][k0 + e];
    }
  };
#define TG_SB() __builtin_amdgcn_sched_barrier(0)
  // tile t: weights in LDS[buf], activations in `a`; meanwhile the weights of tile t+2 are requested into `lb`,
  // those of tile t+1 (`sb`) move to LDS[buf ^ 1], and the activations of tile t+2 are requested into `la`
  auto tile = [&](auto hp_tag, int buf, int t, const ARegs& a, ARegs& la, float4* lb, const float4* sb) {
    constexpr bool HP = decltype(hp_tag)::value;
    const int tl = min(t + 2, nkt - 1);
    BFrag c, n;
    read_b(buf, 0, c);
    const int kb = (HP ? t - nkx : t) * BK + acol, width = HP ? d : xw;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const bool kin = kb + 8 * qq < width;
      const float av[4] = {kin ? a.q[qq].x : 0.f, kin ? a.q[qq].y : 0.f, kin ? a.q[qq].z : 0.f, kin ? a.q[qq].w : 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c.b0[e], acc_r, 0, 0, 0);
        if (qq == 0 && e == 0) read_b(buf, 1, n);
        TG_SB();
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c.b1[e], acc_z, 0, 0, 0);
        if (qq == 0 && e == 1) load_a(tl, la);
        if (qq == 0 && e >= 2) load_b(tl, e - 2, lb);
        if (qq == 1 && e < 2) store_b(buf ^ 1, t + 1, e, sb);
        TG_SB();
        if (HP) acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c.b2[e], acc_hn, 0, 0, 0);
        else acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c.b2[e], acc_in, 0, 0, 0);
        TG_SB();
      }
      if (qq == 0) c = n;
    }
    if (HP && t == nkx + nt) {  // this lane holds part of h[m0.., j0..j0+32): keep it for the epilogue
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int k0 = 8 * (2 * ks + qq) + 4 * fk;
        Hs[arow][k0] = a.q[qq].x; Hs[arow][k0 + 1] = a.q[qq].y;  // (columns past d are never read back)
        Hs[arow][k0 + 2] = a.q[qq].z; Hs[arow][k0 + 3] = a.q[qq].w;
      }
    }
    __syncthreads();
  };
#undef TG_SB
  using HP0 = std::integral_constant<bool, false>;
  using HP1 = std::integral_constant<bool, true>;
  ARegs a0, a1, a2;
  float4 rb0[NBL], rb1[NBL];
  load_a(0, a0);
  load_a(min(1, nkt - 1), a1);
#pragma unroll
  for (int i = 0; i < NBL; ++i) load_b(0, i, rb0);
#pragma unroll
  for (int i = 0; i < NBL; ++i) load_b(min(1, nkt - 1), i, rb1);
#pragma unroll
  for (int i = 0; i < NBL; ++i) store_b(0, 0, i, rb0);
  __syncthreads();
  // straight-line loops over tile TRIPLES would keep the three activation register sets in place; pairs with one
  // rotation copy per tile are simpler and the copies (8 moves) vanish next to 24 MFMAs
  int t = 0;
  auto step = [&](auto hp_tag, int buf, float4* lb, const float4* sb) {
    tile(hp_tag, buf, t, a0, a2, lb, sb);
    a0 = a1;
    a1 = a2;
    ++t;
  };
  while (t + 2 <= nkx) {
    step(HP0{}, 0, rb0, rb1);
    step(HP0{}, 1, rb1, rb0);
  }
  if (t < nkx) {  // odd number of message tiles: the memory tiles start in LDS[1]
    step(HP0{}, 0, rb0, rb1);
    while (t + 2 <= nkt) {
      step(HP1{}, 1, rb1, rb0);
      step(HP1{}, 0, rb0, rb1);
    }
    if (t < nkt) step(HP1{}, 1, rb1, rb0);
  } else {
    while (t + 2 <= nkt) {
      step(HP1{}, 0, rb0, rb1);
      step(HP1{}, 1, rb1, rb0);
    }
    if (t < nkt) step(HP1{}, 0, rb0, rb1);
  }
  if (ks == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red[0][rw][r][lane] = acc_r[r];
      red[1][rw][r][lane] = acc_z[r];
      red[2][rw][r][lane] = acc_in[r];
      red[3][rw][r][lane] = acc_hn[r];
    }
  }
  __syncthreads();
  if (ks == 1) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc_r[r] += red[0][rw][r][lane];
    acc_z[r] += red[1][rw][r][lane];
    acc_in[r] += red[2][rw][r][lane];
    acc_hn[r] += red[3][rw][r][lane];
  }
  const int j = min(j0 + fr, d - 1);
  const bool jok = j0 + fr < d;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lr = rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
    const int64_t m = m0 + lr;
    const float hold = Hs[lr][fr];
    const int64_t orow = orow_s[lr];
    const float rg = fast_sigmoid(acc_r[r] + br);
    const float zg = fast_sigmoid(acc_z[r] + bz);
    const float hn = acc_hn[r] + bhn;
    const float ng = fast_tanh(acc_in[r] + bin + rg * hn);
    if (jok && m < M) {
      g.out[orow * g.ldo + j] = (1.f - zg) * ng + zg * hold;
      if (g.gates) {
        float* gp = g.gates + m * 4 * (int64_t)d + j;
        gp[0] = rg; gp[d] = zg; gp[2 * d] = ng; gp[3 * d] = hn;
      }
    }
  }
}

