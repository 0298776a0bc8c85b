// Dense building blocks on the gfx950 matrix cores: float32-in / float32-accumulate MFMA
// (v_mfma_f32_32x32x2_f32).  The path must match a float32 CPU reference to 1e-4
// relative, so bf16/fp8 MFMA are not an option; the f32 MFMA is bit-for-bit a k-ordered
// fmaf chain.
#pragma once
#include "tg_common.h"

namespace tg {

struct CollateRider;

// one K-slice of the A operand: rows optionally gathered through `idx`
struct ASeg {
  const float* p;
  int64_t ld;          // row stride in floats
  int w;               // width (multiple of 4)
  const int64_t* idx;  // nullable: row m reads table row idx[m]
};

// C[m, n] = act(alpha * (sum_k A[m,k] * W(n,k) + bias[n])), A = [a0 | a1]
struct GemmArgs {
  int64_t m_cap;
  const int32_t* m_dev;  // nullable: live row count on device (<= m_cap)
  int64_t m_hint;        // 0, or the caller's estimate of *m_dev (performance only: sizes the riders of the launch)
  int n, k;
  ASeg a0, a1;
  const float* w;
  int64_t ldw;
  int w_kmajor;  // 0: W[n*ldw + k] (torch Linear)   1: W[k*ldw + n]
  const float* bias;
  float* c;
  int64_t ldc;
  const int32_t* c_rows;     // nullable: output row m is written to row c_rows[m]
  const uint8_t* row_valid;  // nullable: rows with 0 are written as zeros
  float alpha;
  int relu;
  int nbatch;                // batched over blockIdx: per-batch element offsets below
  int64_t a0_bs, w_bs, bias_bs, c_bs;
  // backward-pass epilogues: C = v * (relu_mask[m, n] > 0) (mask shares ldc-style indexing with
  // its own stride), and C += v instead of C = v
  const float* relu_mask;
  int64_t ld_mask;
  int accumulate;
  // bias[n] * bias_rs[m * ld_brs + batch] instead of bias[n] (attention with dropout: the value
  // bias is weighted by the sum of the kept, rescaled probabilities)
  const float* bias_rs;
  int64_t ld_brs;
  // second bias added only on rows with bias2_valid[m] != 0 (fused attention: the constant that the
  // reference zeroes together with the attention output of neighbour-less centres)
  const float* bias2;
  const uint8_t* bias2_valid;
  int dbg;  // diagnostic bits, 0 in production
  // A operand assembled from the stream-K pieces of the producing product (gemm_sk_partials) instead of a0 / a1:
  // A[m, k] = act(ask_alpha * (sum of pieces + ask_bias[k] + ask_valid[m] * ask_bias2[k]))
  const float* ask_part;
  int ask_U, ask_nkt, ask_NT, ask_pieces;
  float ask_rcpU;  // set by gemm_launch
  const float *ask_bias, *ask_bias2;
  const uint8_t* ask_valid;
  int ask_relu;
  float ask_alpha;
  // second, scattered destination of the plain epilogue (nullable): output row m < c2_m with c2_rows[m] >= 0 is ALSO
  // written to row c2_rows[m] of c2.  The write-back rider (tg_common.h: WbRider) uses it to store h(t-) of the winning
  // positions straight into the left memory (STEP 6, tiger.py:253-255).
  float* c2;
  const int32_t* c2_rows;
  int64_t c2_m, ldc2;
  // third and fourth K-slice of A (k_gemm_ks16's second problem only: the raw messages of the split updater are four
  // gathered segments).  a2.p == nullptr with a2.w > 0: a slice of zeros (no edge table, feature_getter.py:95-99)
  ASeg a2, a3;
};

// ---- split updater (streaming with fixed parameters; GRU, raw messages, upd_src = left) --------------------------------
// The eager updater's row of a positive node v, pending[v] = GRU(msg_v, left[v]) (update_modules.py:33-37 once per stored
// message, tiger_hip.h: tg_model.pending_vals), has two halves.  gi = W_ih msg_v + b_ih (80 % of the flops) reads the raw
// message only, which STEP 5 builds from the PRE-batch snapshot (tiger.py:422-442): it does not depend on the attention
// block and runs as a second problem of fc1's launch (k_gemm_ks16, A = four gathered segments).  gh = W_hh h + b_hh needs
// h = left[v] = h(t-) of the winning position = fc2's row = W2 t + b2 (basic_modules.py:16-19), which is linear in fc1's
// output t: with the parameter product W_hh W2 (tg_attn_fuse, the blob's tail) the tail
//   [h | gh_r | gh_z | gh_n] = t [W2 ; W_hh W2]^T + [b2 ; W_hh b2 + b_hh],   K = d
// no longer waits for fc2 - it SHARES fc2's launch (GruTail as a rider of k_gemm_direct_r) and finishes the gates.  The
// step has no updater launch.  h of the tail is bit-identical to fc2's row (same kernel scheme, same k order).
struct GruTail {
  int64_t cap;
  const int32_t* n_dev;     // live rows (the batch's unique positive nodes)
  int d;
  const float* t;           // fc1's output [Q, d]
  const int64_t* t_rows;    // row of t per tail row (the winner's position in cat[src, dst])
  const float *w, *b;       // [4d, d], [4d]
  const float* gi;          // [cap, 3d]
  float* out;               // pending_vals
  const int32_t* out_rows;  // node (row) per tail row
  float* out2;              // nullable: c_table, row = out + add2[node]
  const float* add2;        // nullable: node features
  int64_t rows_hint;        // 0, or the caller's bound on the live rows (performance only: sizes the blocks)
  // direct != 0 (the tail as a launch of its own, BEHIND fc2): the rows are read as the reference reads them - t = the
  // updater-source memory (rows t_rows = node ids), w / b = weight_hh / bias_hh as stored [3d, d]; three planes and h itself
  int direct;
  unsigned blocks;          // set by the launcher
  template <int NS>
  __device__ void run(unsigned bid) const;
};
int gru_tail_launch(const GruTail& t, hipStream_t st);  // the tail as a launch of its own
constexpr int TG_SK_WORKERS = 256;      // default number of workers (one per CU)
constexpr int TG_SK_WORKERS_MAX = 512;  // the workspace is sized for this many (TG_SK_WORKERS env knob)
constexpr size_t TG_SK_WS_FLOATS = (size_t)TG_SK_WORKERS_MAX * 2 * 4096;
// stream-K plan of a product that cannot fill the chip (see tg_gemm.hip)
struct SkPlan {
  int U, nkt, tiles, NT, MT, pieces;  // units per worker, k-tiles per tile, tiles, column tiles, row tiles, max pieces per tile
  float* part;            // [workers][2][64 * 64]
};
// Launches the piece kernel for `g` (bias / activation of g are NOT applied: the consumer applies them) and
// returns true, or returns false (nothing launched) when the shape does not call for it.
bool gemm_sk_partials(const GemmArgs& g, float* ws, size_t ws_floats, hipStream_t st, SkPlan* plan);
// Launches the LDS-free K-split kernel for `g` (bias / activation applied: final values) and returns true, or returns
// false (nothing launched) when the shape does not call for it (tg_gemm.hip: k_gemm_ks16).
// rider (nullable): the write-back rider shares the launch (*rode tells whether)
// second (nullable): another product of the same kind (A of up to four segments) as further persistent blocks of the launch;
// *second_rode tells whether it was taken (only together with the write-back rider)
// fc1 + fc2 of the attention block in one launch (tg_gemm.hip: k_gemm_ks16_fc2); false = not applicable, nothing launched
bool gemm_fc12_launch(const GemmArgs& g, const GemmArgs& g2, hipStream_t st, const WbRider* rider, bool* rode);
bool gemm_ks16_launch(const GemmArgs& g, hipStream_t st, const WbRider* rider = nullptr, bool* rode = nullptr,
                      const GemmArgs* second = nullptr, bool* second_rode = nullptr);

// Riders (nullable, one at most): work that shares the launch as its FIRST workgroups - the one-pass write-back (WbRider,
// tg_common.h) or the collate part of the next batch (CollateRider, tg_sample.h).  Not every kernel hosts them: *rode
// tells the caller whether the rider was launched - otherwise the caller launches that work itself.
// tail (nullable, instead of a rider): the split updater's tail shares the launch (LDS-free short-K products only)
// second (nullable, instead of a rider): a long-K product with A of up to four segments (k_gemm_ks16's blocks) shares it
int gemm_launch(const GemmArgs& g, hipStream_t st, const WbRider* rider = nullptr, bool* rode = nullptr,
                const CollateRider* collate = nullptr, const GruTail* tail = nullptr, const GemmArgs* second = nullptr);

struct GruArgs {
  int64_t cap;
  const int32_t* n_dev;
  int d, xw;
  ASeg x;  // messages  [*, xw]
  ASeg h;  // old memory [*, d]
  const float *w_ih, *w_hh, *b_ih, *b_hh;
  float* out;
  int64_t ldo;
  const int32_t* out_rows;  // nullable
  float* gates;             // nullable [cap, 4, d]: r, z, n, h_n (+bias) per live row, for the backward pass
  // nullable: a second copy of the new rows, dense by launch row m ([cap, d]), plus row add2[node(m), :] when add2 is given,
  // node(m) = the row's output / memory-gather index (the eager updater, where both are the node id, hands the
  // attention-centre form of its rows, h + node features, to the query-row product)
  float* out2;
  const float* add2;
  int out2_by_row;          // out2 is a per-node table (tg_model.c_table): row m goes to its output row, not to row m
  int dbg;                  // diagnostic bits, 0 in production
  int64_t rows_hint;        // upper bound of live rows known on the host (0 = unknown), picks the tile height
  int tail_blocks;          // set by gru_launch: leading blocks that run the 16-column tail (k_gru<3, 4> only)
  // k-tiles [x_skip_at, x_skip_at + x_skip_n) of the message operand are known to be all zero (the edge-feature segment
  // of a raw mailbox row when there is no edge table, feature_getter.py:95-99): they are not loaded and not multiplied
  int x_skip_at, x_skip_n;
};

int gru_launch(const GruArgs& g, hipStream_t st);

// Weight-gradient GEMM: out[n, k] (+)= alpha * sum_m Y[m, n] * X[m, k], X = [x0 | x1] with optional
// row gathers, m < *m_dev (<= m_cap).  The M extent is split over blocks; partial tiles go to
// `part` ([splits, nbatch, N, K] floats, plain stores) and a second launch reduces them in a
// fixed order (deterministic, no float atomics).
struct TnArgs {
  int64_t m_cap;
  const int32_t* m_dev;
  int n, k;
  const float* y;
  int64_t ldy;
  ASeg x0, x1;
  float* out;
  int64_t ldo;
  float alpha;
  int accumulate;
  int nbatch;
  int64_t y_bs, x0_bs, out_bs;
  float* part;
  size_t part_floats;  // capacity of `part`
  // optional column sums of Y from the same pass: bias_out[bz * bias_bs + n] (+)= alpha * sum_m Y[m, n]
  float* bias_out;
  int bias_accumulate;
  int64_t bias_bs;
  const float* bias_rs;  // nullable: row weights, bias_out[n] = sum_m bias_rs[m * ld_brs + brs_col] * Y[m, n]
  int64_t ld_brs;
  int brs_col;           // column of bias_rs used by batch 0 (batch b uses brs_col + b)
};
int gemm_tn_launch(const TnArgs& a, hipStream_t st);
// up to TN_GROUP_MAX independent products in one launch + one reduction launch; `part` is carved per problem
constexpr int TN_GROUP_MAX = 12;
struct TnGroup {
  int n;
  int first_block[TN_GROUP_MAX + 1];
  int first_rblock[TN_GROUP_MAX + 1];
  int splits[TN_GROUP_MAX];
  TnArgs a[TN_GROUP_MAX];
};
int gemm_tn_group_launch(const TnArgs* list, int n, float* part, size_t part_floats, hipStream_t st);
// out[n] (+)= alpha * sum_m Y[m, n]
int colsum_launch(int64_t m_cap, const int32_t* m_dev, int n, const float* y, int64_t ldy, float alpha, float* out,
                  int accumulate, float* part, size_t part_floats, hipStream_t st);

}  // namespace tg
