// Layout of the "tile form" of the pre-multiplied attention weights (tg_attn_fuse appends it to the fused blob) and of
// the LDS image of k_attn_tile (tg_attn_tile.hip): the whole attention block of temporal_agg_modules.py:48-81,210-235
// - G product, gather / softmax core, merged value-out-fc1 product, fc2 - for a tile of 16 centres in ONE workgroup,
// with G and S living in LDS only.  Not part of the C ABI.
#pragma once
#include "tg_common.h"

namespace tg {

constexpr int TILE_M = 16;  // centres per workgroup = rows of v_mfma_f32_16x16x4_f32

// Weights are stored FRAGMENT-MAJOR: a weight W[N][K] (out = A W^T) is cut into (column tile nt of 16 outputs, k-chunk kc
// of 16 inputs) blocks of 256 floats; inside a block lane l of the wavefront owns floats [4 l, 4 l + 4) =
// W[16 nt + l % 16][16 kc + 4 (l / 16) + j], j = 0..3 - exactly the four B operands lane l feeds to the four MFMA steps
// of that chunk, so a wavefront fetches a block as ONE fully coalesced 1 KB load.  N and K are padded to multiples of 16
// with zeros.  The k order inside a chunk is permuted between MFMA steps; the sum over k does not care.
struct TileDims {
  int d, de, nh, kvw, nk;  // de = 0 without an edge table (compact fused weights: key rows have no edge segment)
  int d_p, nk_p;           // padded to 16
  int KCd, KCnk, NTg, NTd; // k-chunks of d / nk, column tiles of the G product / of the d-wide products
  int gs_ld, c_ld;         // LDS row strides in floats (multiples of 64: the XOR swizzle permutes float4 chunks in groups of 16)
  size_t o_wqk, o_gconst, o_w1f, o_b1, o_c1, o_w2, o_b2, floats;  // offsets (floats) inside the tile section
};

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

inline TileDims tile_dims(const tg_model* m) {
  TileDims t{};
  t.d = m->d;
  t.de = m->efeats ? m->d_e : 0;
  t.nh = m->n_head;
  t.kvw = 2 * t.d + t.de;
  t.nk = t.nh * t.kvw;
  t.d_p = rup(t.d, 16);
  t.nk_p = rup(t.nk, 16);
  t.KCd = t.d_p / 16;
  t.KCnk = t.nk_p / 16;
  t.NTg = t.nk_p / 16;
  t.NTd = t.d_p / 16;
  t.gs_ld = rup(t.nk_p, 64);
  t.c_ld = rup(t.d_p, 64);
  size_t o = 0;
  t.o_wqk = o;    o += (size_t)t.NTg * t.KCd * 256;
  t.o_gconst = o; o += t.nk_p;
  t.o_w1f = o;    o += (size_t)t.NTd * (t.KCnk + t.KCd) * 256;
  t.o_b1 = o;     o += t.d_p;
  t.o_c1 = o;     o += t.d_p;
  t.o_w2 = o;     o += (size_t)t.NTd * t.KCd * 256;
  t.o_b2 = o;     o += t.d_p;
  t.floats = o;
  return t;
}

// LDS bytes of a k_attn_tile workgroup of nwv wavefronts: G/S tile, centre tile, fc1-output tile, tail partials, flags
inline size_t tile_lds_bytes(const TileDims& t, int nwv) {
  return ((size_t)TILE_M * t.gs_ld + 2 * (size_t)TILE_M * t.c_ld + (size_t)nwv * 2 * 256 + 64) * sizeof(float);
}
constexpr size_t TILE_LDS_MAX = 160 * 1024;

// non-zero when the tile kernel can run a model of these dimensions, 0 = the tile form does
// not apply (rows wider than a wavefront's 64 x 4 columns, or tiles beyond one CU's LDS)
inline int tile_waves_for_shape(const tg_model* m) {
  if (m->d <= 0 || (m->d % 4) || (m->d_e % 4) || m->d > 256 || m->d_e > 256 || m->n_neighbors > TG_WAVE) return 0;
  if (m->n_head != 1 && m->n_head != 2 && m->n_head != 4) return 0;
  const TileDims t = tile_dims(m);
  return tile_lds_bytes(t, 12) <= TILE_LDS_MAX ? 12 : 0;  // the larger of the two workgroup shapes (12 and 8 wavefronts)
}
inline int tile_waves(const tg_model* m) { return m->attn_fused ? tile_waves_for_shape(m) : 0; }

size_t attn_fused_floats_of(size_t d, size_t nk);  // tg_fuse.hip: size of the row-major section that precedes the tile section
int attn_tile_prepare();                           // tg_attn_tile.hip: LDS limits of the kernels (outside any stream capture)
int attn_tile_applies(const tg_model* m);

}  // namespace tg
