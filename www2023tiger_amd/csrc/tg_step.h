// Internal layout of the fused step's workspace and the pieces of tg_stream_step that the
// training step (tg_train.hip) re-uses.  Not part of the C ABI.
#pragma once
#include "tg_dense.h"

struct tg_profiler;

namespace tg {

static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct Carver {
  char* p;
  size_t left;
  bool ok = true;
  Carver(void* ws, size_t bytes) : p((char*)ws), left(bytes) {}
  template <typename T>
  T* take(size_t count) {
    const size_t b = align16(count * sizeof(T));
    if (b > left || !p) {
      ok = false;
      return nullptr;
    }
    T* r = (T*)p;
    p += b;
    left -= b;
    return r;
  }
};

// intermediates of the temporal attention; all of them survive until the end of the step,
// which is what the backward pass reads
struct AttnWs {
  float *cc, *qp, *g, *s, *o, *hh, *t, *qconst;
  float* rsum;  // [Q, n_head] sum of the kept, rescaled attention probabilities (training with dropout)
  uint8_t* valid;
  float* sk;  // stream-K partial tiles of the merged fc1 product (TG_SK_WS_FLOATS)
};

struct StepWs {
  uint8_t* flags;            // involved byte flags        (zeroed every step)
  unsigned long long* best;  // per involved rank          (zeroed every step)
  int32_t* counts;           // [8]: involved, outdated, unique positives, restarted | batch-min-time key, 3 spare
                             //                            (zeroed every step)
  size_t zero_bytes;         // size of the contiguous zeroed region starting at flags
  uint64_t* bm;              // involved bitmap, packed from the flags
  uint32_t *rank, *rank_out;
  int64_t *nids3, *eids, *involved, *outdated, *upos, *index;
  double* ts3;
  float *ts3f, *l1_ts, *reprs;
  int64_t *l1_nids, *l1_eids;
  int32_t* out_pos;
  int32_t* upos32;
  int32_t* win_row;  // [2B] per position of cat[src, dst]: its node if the position wins the node's dedup, else -1 (write-back rider)
  bool wb_rode;      // STEP 4-6 rode on the launch of the attention block's last product: no write-back launch
  float *snap, *snap_ts;  // one-launch write-back: message-source rows (+ node features) / times of cat[src, dst], pre-batch
  void* scan_ws;
  size_t scan_bytes;
  AttnWs attn;
  void* apply_ws;
  size_t apply_bytes;
  // resolved per call (io overrides)
  int64_t *l1n, *l1e, *inv;
  float* l1t;
  bool dedup_done;  // the positive-node dedup already ran inside the forward launches
  bool eager;       // STEP 1-2 gathered precomputed updater rows; the updater runs at the end of the step instead
  unsigned long long* best_id;  // [n_nodes] dedup slots of a lean step (indexed by node id; zero between steps)
  bool lean;        // no involved / outdated sets are formed (tg_step_io.lean)
  bool fused_wb;    // STEP 4-6 run as one launch (needs the snapshot taken by the direct centres launch)
  bool direct;      // ... and no compact copy of the involved rows was made (centres / neighbours read the tables)
  bool gtab;        // the folded queries come from the per-node table; the step refreshes its positive nodes' rows at the end
  hipEvent_t collate_done;  // nullable, the caller's: recorded right behind the first-hop sampler (the batch's id list exists)
  bool collate_recorded;
  bool prefetch;    // the step runs the collate part of the NEXT batch on its last launch (tg_step_io.prefetch_state)
  bool prefetch_side;  // ... large batch: its sampler half on the side lane beside the updater, its centres half behind the query rows
  PosArgs pos_args;   // the step's dedup arguments / direct-centres arguments (the prefetch builds the next batch's
  DirectArgs da_args; // centres rider from them at the end of the step)
  // --n_layers 2: second-hop lists of the Q*K neighbour slots, the slots' query times (the roots'), their embeddings
  int64_t *h2n, *h2e;
  float *h2t, *ts2, *emb2;
  AttnWs attn2;
  // split updater (tg_dense.h: GruTail): time segments of the 2B snapshot positions, the winners' other-endpoint positions
  // and edge ids, W_ih msg + b_ih of the winners
  float *snap_te, *gi;
  int64_t *oth, *weid;
  bool upd_done;  // the eager updater's rows of this batch were finished inside the attention block's launches
  bool sampler_rode;  // collate prefetch: the next batch's sampler rode on fc2's launch already (its centres follow on the last)
  const WbRider* ext_rider;  // tg_part_step: the planned write-back's first launch rides on the embedding step's fc1 / fc2
  bool tail_pending;  // ... or their input-side product was (on fc2's launch): the tail is launched where the updater was
  GruTail tail;
};
// one step's plan of the split updater: the second problem of fc1's launch and the tail on fc2's (attn_forward_fused)
struct GruSplit {
  GemmArgs gi;
  GruTail tail;
  int variant;   // 1: gi on fc1's launch + tail on fc2's (pre-multiplied W_hh W2); 2: gi on fc2's launch + the tail behind it
  bool gi_done;  // variant 2: gi rode on fc2's launch
  bool done;     // variant 1: the rows are finished
};
const float* gru_tail_weights(const tg_model* m);
// tg_stream_step with a rider handed in by the caller (tg_part_step; *rode: whether a launch of the step hosted it)
int stream_step_ext(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, void* ws, size_t ws_bytes, hipStream_t st,
                    const WbRider* ext_rider, bool* rode);
// the eager updater over a list of state rows (tg_part_step): pending[rows32[i]] = updater(upd memory, mailbox)[rows[i]]
int apply_messages_rows(const tg_model* m, const int64_t* rows, const int32_t* rows32, const int32_t* n_dev, int64_t cap,
                        uint32_t* err, void* ws, size_t ws_bytes, hipStream_t st);  // the tail of the tg_attn_fuse blob, or nullptr (tg_fuse.hip)

bool carve_step(const tg_model* m, int64_t B, Carver& cv, StepWs& w, int n_layers = 1);
int attn_dims_ok(const tg_model* m);
int gtab_rows(const tg_model* m, int64_t cap, const int64_t* nids, const int32_t* rows32, const int32_t* n_dev, float* crows,
              hipStream_t st, bool crows_ready, const CollateRider* collate = nullptr, bool* rode = nullptr,
              int64_t rows_hint = 0);
// collate + STEP 1-3 (+ io->h_new); `gates` (nullable) receives the GRU gate activations
// eager: take the outdated nodes' rows from m->pending_vals instead of running the updater (tg_stream_step only)
int step_forward(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, StepWs& w, float* gates, hipStream_t st,
                 tg_profiler* pf, const DropCfg* drop = nullptr, bool eager = false);
// positive-node dedup + STEP 4/5 + restarter targets
int step_writeback_a(const tg_model* m, const tg_step_io* io, StepWs& w, hipStream_t st, tg_profiler* pf);
// STEP 6 + workspace clean-up + offset advance
int step_writeback_b(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, StepWs& w, hipStream_t st, tg_profiler* pf);

// a second stream beside the caller's (tg_train.hip: train_lane) with a pool of events: `after_main` makes the lane wait for
// everything enqueued on the main stream so far (under capture: a dependency edge of the graph)
struct SideCtx {
  hipStream_t s;
  hipEvent_t ev[16];
  int n, used;
  bool after_main(hipStream_t st) {
    if (used >= n) return false;
    const bool ok = hipEventRecord(ev[used], st) == hipSuccess && hipStreamWaitEvent(s, ev[used], 0) == hipSuccess;
    ++used;
    return ok;
  }
};
// mutual-learning half of the training step (tg_restart.hip)
size_t mutual_ws_bytes(const tg_model* m, const tg_seq_restarter* r, int64_t B);
int mutual_step(const tg_model* m, const tg_tcsr* g, const tg_step_io* sio, const StepWs& sw,
                const tg_seq_restarter* r, const tg_seq_restarter* gr, const float* st_left, const float* st_right,
                float* g_left, float* g_right, float* loss_out, int32_t* flag_out, float* part, size_t part_floats,
                void* ws, size_t ws_bytes, const DropCfg& dc, hipStream_t st, int phase = 3, SideCtx* side = nullptr);

}  // namespace tg
