// EXPERIMENT, not part of libtiger_hip.so: the GRU cell with direct-to-LDS operand staging
// (global_load_lds_dwordx4) instead of global -> registers -> ds_write.  It was dropped into tg_gemm.hip next to
// k_gru (same GruArgs, same block -> tile map, selected by an environment knob) and passed the parity tests of
// tests/test_hip_parity.py (GRU operator, C2 full size, 16-batch soak).  Measured on MI355X at C2 (rocprofv3):
//   k_gru<4,2> (register staging)            60.4 us      ablated: no loads / stores / barrier  48.0 us
//   k_gru_dl, ds_read2_b32 fragments         64.8 us      (4-way bank conflicts: 32-bank addressing)
//   k_gru_dl, ds_read_b64 fragments          61.0 us      without its loads 52.7 us
//   k_gru_dl, ds_read_b128 quads (this file) 60.7 us      without its loads 50.3 us; 4 LDS buffers: 60.9 us
// i.e. the transfers cost ~10 us however they are issued (see also gru_activations_in_registers.hip: halving the
// LDS writes changes nothing either), so the direct path only saves registers here.  Kept for the next round
// (larger tiles per staged byte).
// ---------------------------------------------------------------------------------
// GRU cell, direct-to-LDS staging (global_load_lds_dwordx4): the operand tiles travel from global memory
// into LDS without passing through registers - no staging VGPRs, no ds_write, no tail selects in the loop
// (the ablation of k_gru puts loads + LDS stores + barrier at 12 of its 60 us at C2).
//  * a wavefront-wide direct load writes 1 KB of CONSECUTIVE LDS (lane i -> base + 16 i), so a tile is stored
//    row-major without padding, 8 lanes per 128-byte row, 8 rows per instruction; bank conflicts of the MFMA
//    operand reads are avoided by an XOR swizzle instead of padding: lane (row, p) fetches the row's 16-byte
//    chunk p ^ ((row >> 1) & 7), i.e. chunk c of a row lives at position c ^ ((row >> 1) & 7);
//  * the operand fragments are read with ds_read_b128, one chunk per lane feeding four MFMA k-steps (see below);
//  * chunks past the end of a segment (message width, memory width) are fetched from a 16-byte zero;
//  * three LDS tile buffers: the loads of tile t+2 are issued during tile t and waited for (counted vmcnt)
//    before the barrier that ends tile t+1.  Every wavefront issues exactly four loads per tile (28 row groups
//    + 4 dummies) so that the count is uniform.
// ---------------------------------------------------------------------------------
__device__ float4 g_zero16;  // zero-initialised

template <int ABL = 0>  // diagnostic: bit 0 drops the loop's loads (garbage results, shows the cost of the rest)
__global__ void __launch_bounds__(512) k_gru_dl(GruArgs g) {
  constexpr int NW = 4, THREADS = 512, BM = 128;  // two k-groups of NW row waves
  constexpr int TILE = 32 * 256;   // floats per tile buffer: 32 row groups of 8 rows x 32 floats
  constexpr int BOFF = 16 * 256;   // the weight rows follow the 128 activation rows
  constexpr int NBUF = 4;          // LDS tile buffers: the loads of tile t + NBUF - 1 are issued during tile t
  __shared__ __attribute__((aligned(16))) float tiles[NBUF * TILE];
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
  // ---- this lane's four load slots: row group (wave + 8 slot), row rl of it, chunk position p
  const int rl = lane >> 3, p = lane & 7;
  const float* px[4];  // source row during the message tiles
  const float* ph[4];  // source row during the memory tiles
  int cl[4];           // logical chunk fetched into position p
  const float* zero = reinterpret_cast<const float*>(&g_zero16);
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int gi = wave + 8 * sl;  // 0..15 activations, 16..27 weights, 28..31 dummies
    if (gi < 16) {
      const int row = gi * 8 + rl;
      const int64_t m = min(m0 + row, M - 1);
      px[sl] = g.x.p + (g.x.idx ? g.x.idx[m] : m) * g.x.ld;
      ph[sl] = g.h.p + (g.h.idx ? g.h.idx[m] : m) * g.h.ld;
      cl[sl] = (ABL & 2) ? p : p ^ ((row >> 1) & 7);
    } else if (gi < 28) {
      const int L = (gi - 16) * 8 + rl;  // row of the [3 planes x 32] weight tile
      const int jc = min(j0 + (L & 31), d - 1);
      px[sl] = g.w_ih + ((int64_t)(L >> 5) * d + jc) * xw;
      ph[sl] = g.w_hh + ((int64_t)(L >> 5) * d + jc) * d;
      cl[sl] = (ABL & 2) ? p : p ^ ((L >> 1) & 7);
    } else {
      px[sl] = ph[sl] = zero;
      cl[sl] = -1;  // always the zero chunk
    }
  }
  const int nkx = (xw + BK - 1) / BK, nkh = (d + BK - 1) / BK;
  const int nkt = nkx + nkh;
  auto issue = [&](int t, int sl, int bufoff) {  // slot sl of tile t -> LDS
    const bool hp = t >= nkx;
    const int k = (hp ? t - nkx : t) * BK + cl[sl] * 4;
    const int width = hp ? d : xw;
    const float* src = (cl[sl] >= 0 && k < width) ? (hp ? ph[sl] : px[sl]) + k : zero;
    // Inline assembly on purpose: through __builtin_amdgcn_global_load_lds the compiler's wait-count pass treats
    // every later LDS read as possibly aliasing the transfer and puts s_waitcnt vmcnt(0) in front of it, which
    // serialises the loop.  The transfers are ordered by hand instead: counted vmcnt before each barrier.  (The low
    // 32 bits of a generic pointer into LDS are the LDS byte address; M0 carries it, lane i lands at M0 + 16 i.)
    const uint32_t dst = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)&tiles[bufoff + (wave + 8 * sl) * 256]);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");  // M0 is not used by anything else in this kernel
  };
  f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc_r[i] = acc_z[i] = acc_in[i] = acc_hn[i] = 0.f;
  // operand read addresses (floats): row base + swizzled chunk + the half-wave's element pair
  const int swz4 = ((fr >> 1) & 7) << 2;
  // Operand fragments: a whole 16-byte chunk per lane through ONE ds_read_b128.  The half-wave fk = 0 reads
  // chunk 2 j, the half-wave fk = 1 chunk 2 j + 1, and the four floats feed four consecutive MFMA k-steps -
  // a permutation of the k order inside every group of eight columns, applied to both operands alike, that
  // uses every byte read.  With the swizzle above each 16-lane read group touches 16 different 16-byte slots
  // of the 256-byte bank row: conflict-free.  Inline assembly, because loads written in C++ are narrowed by
  // the optimiser into dword loads and re-paired as ds_read2_b32 (32-bank addressing, 4-way conflicts here);
  // the compiler does not count these reads in lgkmcnt, so `landed` waits for them by hand and, by naming the
  // registers as in/out operands, keeps every consumer behind the wait.
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  struct Frag {
    f32x4v a, b0, b1, b2;
  };
  const unsigned lds0 = (unsigned)(uintptr_t)&tiles[0];
  const int lc = (fk << 2) ^ swz4;  // chunk offset (floats) of quad j is (8 j) ^ lc
  const unsigned a_addr = lds0 + 4u * (unsigned)((rw * 32 + fr) * 32);
  const unsigned b_addr = lds0 + 4u * (unsigned)(BOFF + fr * 32);
  auto read_quad = [&](int bufoff, int j, Frag& f) {  // columns 8 j .. 8 j + 7 of the tile in LDS
    const unsigned o = 4u * (unsigned)(bufoff + ((8 * j) ^ lc));
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.a) : "v"(a_addr + o) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.b0) : "v"(b_addr + o) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.b1) : "v"(b_addr + o) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(f.b2) : "v"(b_addr + o) : "memory");
  };
  auto landed = [&](Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2));
  };
#define TG_SB() __builtin_amdgcn_sched_barrier(0)
  auto tile = [&](auto hp_tag, int t, int cur, int nxt2) {
    constexpr bool HP = decltype(hp_tag)::value;
    const int tl = min(t + NBUF - 1, nkt - 1);  // past the end: a redundant reload keeps the count uniform
    Frag c, n;
    read_quad(cur, ks * 2, c);
    landed(c);
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a[e], c.b0[e], acc_r, 0, 0, 0);
        if (qq == 0 && e == 0) read_quad(cur, ks * 2 + 1, n);
        TG_SB();
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a[e], c.b1[e], acc_z, 0, 0, 0);
        if (!(ABL & 1) && (e & 1)) issue(tl, qq * 2 + (e >> 1), nxt2);
        TG_SB();
        if (HP) acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a[e], c.b2[e], acc_hn, 0, 0, 0);
        else acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a[e], c.b2[e], acc_in, 0, 0, 0);
        TG_SB();
      }
      if (qq == 0) {
        landed(n);
        c = n;
      }
    }
    if (HP && t == nkx + nt) {  // this activation tile is h[m0.., j0..j0+32): keep it for the epilogue
      for (int f = tid; f < BM * 32; f += THREADS) {
        const int row = f >> 5, kk = f & 31;
        Hs[row][kk] = tiles[cur + row * 32 + ((((kk >> 2) ^ ((row >> 1) & 7)) << 2) | (kk & 3))];
      }
    }
    // the loads of tile t + 1 must have landed; those of the NBUF - 2 tiles after it may still travel
    __builtin_amdgcn_s_waitcnt(0x0F70 | (4 * (NBUF - 2)));  // vmcnt(4 (NBUF - 2)); lgkmcnt / expcnt at "no wait"
    __syncthreads();
  };
#undef TG_SB
  using HP0 = std::integral_constant<bool, false>;
  using HP1 = std::integral_constant<bool, true>;
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b)
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) issue(min(b, nkt - 1), sl, b * TILE);
  __builtin_amdgcn_s_waitcnt(0x0F70 | (4 * (NBUF - 2)));  // tile 0 has landed
  __syncthreads();
  int cur = 0, last = (NBUF - 1) * TILE;  // buffer of tile t, buffer that tile t + NBUF - 1 goes to
  auto rotate = [&]() {
    last = cur;
    cur = cur + TILE == NBUF * TILE ? 0 : cur + TILE;
  };
  int t = 0;
  for (; t < nkx; ++t) {
    tile(HP0{}, t, cur, last);
    rotate();
  }
  for (; t < nkt; ++t) {
    tile(HP1{}, t, cur, last);
    rotate();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // the redundant tail loads still target `tiles`
  __syncthreads();
  float* red = tiles;  // [4][NW][16][64]: the tile buffers are dead now
  if (ks == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red[((0 * NW + rw) * 16 + r) * 64 + lane] = acc_r[r];
      red[((1 * NW + rw) * 16 + r) * 64 + lane] = acc_z[r];
      red[((2 * NW + rw) * 16 + r) * 64 + lane] = acc_in[r];
      red[((3 * NW + rw) * 16 + r) * 64 + lane] = acc_hn[r];
    }
  }
  __syncthreads();
  if (ks == 1) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc_r[r] += red[((0 * NW + rw) * 16 + r) * 64 + lane];
    acc_z[r] += red[((1 * NW + rw) * 16 + r) * 64 + lane];
    acc_in[r] += red[((2 * NW + rw) * 16 + r) * 64 + lane];
    acc_hn[r] += red[((3 * NW + rw) * 16 + r) * 64 + lane];
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

