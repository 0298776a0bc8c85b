/*
 * tiger_hip.h - C ABI of libtiger_hip.so, the MI355X (gfx950) engine for the
 * TIGER event-batch hot path (temporal neighbour sampling -> mailbox consume +
 * GRU -> temporal-attention embedding -> memory / mailbox write-back -> restart).
 *
 * The reference (yzhang1918/www2023tiger @ v1.0.1) has no FFI: its boundary is
 * the Python class API of tiger.data.graph / tiger.model.*.  Each entry point
 * below names the reference code (file:line, relative to the reference root) whose
 * body it replaces; www2023tiger_amd/ keeps the reference's Python signatures and
 * binds these symbols with ctypes.
 *
 * Conventions
 *  - plain C: pointers + sizes; no C++ types, no exceptions cross the boundary.
 *  - every array argument is a raw DEVICE pointer into caller-owned memory unless
 *    its name ends in _host; the library never allocates or frees user-visible
 *    memory - scratch comes from the caller-provided workspace `ws` (16-byte
 *    aligned) whose size the matching *_workspace_bytes() call returns.
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it
 *    and performs no host synchronisation (safe to capture into a hipGraph).
 *  - return value: TG_OK or a negative TG_E* code for argument/shape errors.
 *    Data-dependent invariants (the reference's ValueErrors) are
 *    checked on device and OR-ed into the caller's `err` word (TG_ERR_* bits),
 *    which the Python shim reads when it needs the reference's exception.
 *  - ids are int64 at the boundary (the reference's dtype); timestamps are float64
 *    on the sampler side and float32 on the model side, exactly as in the reference.
 *  - all feature widths (d, d_e) must be multiples of 4 (16-byte rows).
 */
#ifndef TIGER_HIP_H
#define TIGER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 9

/* status codes */
#define TG_OK 0
#define TG_EINVAL (-1)      /* bad argument / shape */
#define TG_EUNSUPPORTED (-2) /* valid in the reference, not built here */
#define TG_EWORKSPACE (-3)  /* workspace too small */
#define TG_EHIP (-4)        /* a HIP runtime call failed (tg_last_hip_error) */

/* device-side invariant bits OR-ed into *err (reference exception in brackets) */
#define TG_ERR_PAST_MEMORY 1u      /* memory.py:45-46  'You are not allowed to modify past memory.' */
#define TG_ERR_DUPLICATE_IDS 2u    /* memory.py:47-48  'Duplicate node ids are not allowed.' */
#define TG_ERR_UNUSED_MESSAGE 4u   /* memory.py:85-87  'Node #n has unused messages.' */
#define TG_ERR_MSG_BEFORE_MEM 8u   /* message_modules.py:158-159 */
#define TG_ERR_MSG_TS_MISMATCH 16u /* tiger.py:325-327 (msg_src == left) */
#define TG_ERR_EVENT_BEFORE_MEM 32u /* tiger.py:437-438 */
#define TG_ERR_XCHG_TIMEOUT 64u    /* tg_part_step: a peer's rows did not arrive within the bounded wait (no reference counterpart) */

int tg_abi_version(void);
/* text of the last HIP runtime error seen by this thread ("" if none) */
const char* tg_last_hip_error(void);

/* ------------------------------------------------------------------------- */
/* Temporal CSR (replaces Graph.__init__/from_data/data2adjlist,              */
/* tiger/data/graph.py:11-42,226-241)                                         */
/* ------------------------------------------------------------------------- */
typedef struct tg_tcsr {
  int64_t num_node;      /* max id + 1; id 0 is the padding node */
  int64_t num_entry;     /* 2 * num_events */
  const int64_t* indptr; /* [num_node + 1] */
  const double* ts;      /* [num_entry]  float64 event time, ascending per node */
  const int32_t* nbr;    /* [num_entry]  the other endpoint */
  const int32_t* eid;    /* [num_entry]  edge id in bits 0..30, bit 31 = "owner is dst" flag */
} tg_tcsr;

/* Host-side build (initialisation only): stable per-node sort by time of the
 * 2E (neighbour, eid, t, flag) entries.  All pointers are HOST pointers.
 * Returns TG_EINVAL if an id is negative / >= num_node or an eid needs more than 31 bits. */
int tg_tcsr_build_host(int64_t num_events, const int64_t* src_host, const int64_t* dst_host,
                       const double* ts_host, const int64_t* eid_host, int64_t num_node,
                       int64_t* indptr_host, double* ts_out_host, int32_t* nbr_out_host,
                       int32_t* eid_out_host);

/* RandEdgeSampler.sample(1) called `count` times (data_loader.py:291-294): per event one
 * randint(0, n_src) then one randint(0, n_dst) on a numpy legacy RandomState.  mt_state: HOST
 * uint32[625] = MT19937 key + position (numpy get_state()[1:3]), advanced in place. */
int tg_rand_edge_pairs_host(uint32_t* mt_state, int64_t n_src, int64_t n_dst, int64_t count,
                            int64_t* src_idx_host, int64_t* dst_idx_host);

/* The same draws on DEVICE (one wavefront, the accept / reject decisions of a state block taken 64 words at a time):
 * mt_state is device uint32[625], advanced in place exactly as the host routine advances it, so host and device
 * draws can continue each other's stream.  src_list / dst_list (device int64, nullable) map the drawn indices to
 * node ids as RandEdgeSampler.sample does (data_loader.py:293-294); out_* are device int64[count]. */
int tg_rand_edge_pairs(uint32_t* mt_state, int64_t n_src, int64_t n_dst, int64_t count, const int64_t* src_list,
                       const int64_t* dst_list, int64_t* out_src, int64_t* out_dst, void* stream);

/* The same build on device for a TIME-ORDERED stream (ts non-decreasing - the caller checks; every
 * JODIE file is): a stable radix sort of the 2E (owner, entry) pairs on the owner id.  All pointers
 * are DEVICE pointers; ids must lie in [0, num_node), eids in [0, 2^31), 2E < 2^32. */
size_t tg_tcsr_build_device_workspace_bytes(int64_t num_events, int64_t num_node);
int tg_tcsr_build_device(int64_t num_events, const int64_t* src, const int64_t* dst, const double* ts,
                         const int64_t* eid, int64_t num_node, int64_t* indptr, double* ts_out,
                         int32_t* nbr_out, int32_t* eid_out, void* ws, size_t ws_bytes, void* stream);

/* Graph.sample_temporal_neighbor(strategy='recent_edges') and Graph.get_history
 * (graph.py:67-127,150-155): per query the last K entries with ts < t (strict),
 * left padded with zeros.  out_dir may be NULL.  If mark_flags != NULL every query
 * id and every sampled neighbour id (padding 0 included) is also flagged in that
 * byte array (uint8[tg_flag_bytes(n_nodes)], zeroed by the caller; plain stores, no
 * atomics) - the fusion of GraphCollator.collate_memory_nodes' set.update
 * (data_loader.py:109-113); tg_unique_compact packs the flags into the node bitmap. */
int tg_sample_recent_edges(const tg_tcsr* g, int64_t n_query, const int64_t* nids, const double* ts,
                           int32_t K, int64_t* out_nbr, int64_t* out_eid, float* out_ts,
                           int64_t* out_dir, uint8_t* mark_flags, void* stream);

/* strategy='recent_nodes' (graph.py:129-143): last occurrence of each distinct
 * neighbour, most recent K of those, ascending in time, left padded. */
int tg_sample_recent_nodes(const tg_tcsr* g, int64_t n_query, const int64_t* nids, const double* ts,
                           int32_t K, int64_t* out_nbr, int64_t* out_eid, float* out_ts,
                           int64_t* out_dir, void* stream);

/* strategy='uniform' (graph.py:101-115): K draws of numpy's legacy
 * RandomState.randint(0, len, K) per non-empty query IN QUERY ORDER, sorted by time.
 * `mt_state` is the 624-word MT19937 key followed by the position word (625 uint32,
 * device memory), advanced in place so consecutive calls continue the stream. */
int tg_sample_uniform(const tg_tcsr* g, int64_t n_query, const int64_t* nids, const double* ts,
                      int32_t K, uint32_t* mt_state, int64_t* out_nbr, int64_t* out_eid, float* out_ts,
                      int64_t* out_dir, void* stream);

/* GraphCollator.check_in_window's comparison (data_loader.py:61-67):
 * out[b,k] = (center[b] == nbr[b,k]) as float32. */
int tg_hits(int64_t B, int32_t K, const int64_t* center, const int64_t* nbr, float* out, void* stream);

/* anonymized_reindex (tiger/model/utils.py:19-27) on an [n, H] id matrix. */
int tg_anonymized_reindex(int64_t n, int32_t H, const int64_t* hist_nids, int64_t* out, void* stream);

/* ------------------------------------------------------------------------- */
/* Sorted-unique compaction on an n-node bitmap                               */
/* ------------------------------------------------------------------------- */
/* number of uint64 words of a bitmap over n_nodes ids */
int64_t tg_bitmap_words(int64_t n_nodes);
int tg_bitmap_mark(int64_t n, const int64_t* ids, uint64_t* bitmap, int64_t n_nodes, void* stream);
/* byte-flag form of the same set (one uint8 per node, padded to a multiple of 64) */
int64_t tg_flag_bytes(int64_t n_nodes);
int tg_flags_mark(int64_t n, const int64_t* ids, uint8_t* flags, int64_t n_nodes, void* stream);
/* If flags != NULL they are first packed into `bitmap` (which is then an output).
 * Reads the bitmap back as the sorted id list (replaces np.sort(list(set)),
 * data_loader.py:121, and the dense local_index of data_classes.py:163-165):
 *   rank[w]    = number of set bits in words [0, w)          (uint32[words + 1])
 *   out_ids[r] = r-th smallest set id                        (capacity `cap`)
 *   *out_count = number of set bits
 * local index of id i  =  rank[i>>6] + popcount(bitmap[i>>6] & ((1<<(i&63))-1)).
 * If and_bitmap != NULL a second list is produced for (bitmap & and_bitmap):
 * and_rank / and_ids / and_count and, per element, its position in the first list
 * (and_pos) - this is the "outdated = involved ∩ has-message" set of
 * MessageStoreNoGradLastOnly.get_outdated_node_ids (memory.py:108-126). */
size_t tg_unique_compact_workspace_bytes(int64_t n_nodes);
int tg_unique_compact(const uint8_t* flags, uint64_t* bitmap, int64_t n_nodes, uint32_t* rank, int64_t* out_ids,
                      int32_t* out_count, int64_t cap, const uint64_t* and_bitmap, uint32_t* and_rank,
                      int64_t* and_ids, int32_t* and_pos, int32_t* and_count, void* ws, size_t ws_bytes,
                      void* stream);

/* select_latest_nids (tiger/model/utils.py:10-16): sorted unique ids of nids[0..n)
 * and, per id, the position of its maximum timestamp, FIRST position among ties.
 * ts_is_f64 selects const double* / const float* for `ts`.  out_* have capacity n. */
size_t tg_select_latest_workspace_bytes(int64_t n, int64_t n_nodes);
int tg_select_latest(int64_t n, const int64_t* nids, const void* ts, int32_t ts_is_f64, int64_t n_nodes,
                     int64_t* out_unique, int64_t* out_index, int32_t* out_count, void* ws,
                     size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Model description shared by the dense / memory entry points               */
/* ------------------------------------------------------------------------- */
typedef struct tg_linear {   /* torch.nn.Linear: y = x W^T + b, W [out, in] row-major */
  const float* w;
  const float* b;
} tg_linear;

enum { TG_SRC_LEFT = 0, TG_SRC_RIGHT = 1 };
enum { TG_TSFM_ID = 0, TG_TSFM_LINEAR = 1, TG_TSFM_MLP = 2 };
enum { TG_UPD_GRU = 0, TG_UPD_MERGE = 1 };

typedef struct tg_model {
  /* sizes */
  int64_t n_nodes;
  int32_t d;        /* memory = node-feature = time-encoding width (tiger.py:58-61) */
  int32_t d_e;      /* edge-feature width (== d when there is no edge table)     */
  int32_t n_neighbors;
  int32_t n_head;
  int32_t msg_src;  /* TG_SRC_*  (tiger.py:87) */
  int32_t upd_src;  /* TG_SRC_*  (tiger.py:88) */
  int32_t tsfm;     /* TG_TSFM_* (tiger.py:109-117) */
  int32_t upd_fn;   /* TG_UPD_*  (tiger.py:120-125) */
  /* state: Memory (memory.py:12-52) x2, mailbox MessageStoreNoGradLastOnly (memory.py:55-138) */
  float* left_vals;   /* [n_nodes, d] */
  float* left_ts;     /* [n_nodes]    */
  uint8_t* left_active;
  float* right_vals;
  float* right_ts;
  uint8_t* right_active;
  float* msg_vals;    /* [n_nodes, 3d + d_e] */
  float* msg_ts;      /* [n_nodes] */
  uint64_t* has_msg;  /* bitmap over n_nodes: replaces the Python set nodes_with_messages */
  /* raw feature tables (feature_getter.py:25-106); NULL => zeros */
  const float* nfeats; /* [n_nodes, d]   */
  const float* efeats; /* [n_edges+1, d_e] */
  /* TimeEncode (time_encoding.py:13-14) */
  const float* te_freq;
  const float* te_phase;
  /* message transform (message_modules.py:29-55): fn.1 and (mlp) fn.4 */
  tg_linear tsfm1, tsfm2;
  /* GRUCell (update_modules.py:33): weight_ih [3d, msg], weight_hh [3d, d] */
  const float *gru_w_ih, *gru_w_hh, *gru_b_ih, *gru_b_hh;
  /* MergeUpdater (update_modules.py:43): MergeLayer fc1 [d, msg+d], fc2 [d, d] */
  tg_linear upd_fc1, upd_fc2;
  /* TemporalAttention (temporal_agg_modules.py:196-235) */
  const float *attn_wq, *attn_wk, *attn_wv; /* [2d,2d], [2d,2d+d_e], [2d,2d+d_e] */
  const float* attn_b_in;                   /* [6d] */
  tg_linear attn_out;                       /* [2d,2d] */
  tg_linear attn_fc1, attn_fc2;             /* merger: [d,3d], [d,d] */
  /* Optional (NULL = off): weights pre-multiplied by tg_attn_fuse for inference with FIXED parameters.
   * The forward pass then runs three products instead of six (see tg_attn_fuse). */
  const float* attn_fused;
  /* Optional (NULL = off): EAGER updates for streaming with FIXED parameters.  pending_vals [n_nodes, d] holds,
   * for every node with a pending message (has_msg bit set), the row the updater would produce when the
   * message is consumed: pending[v] = updater(upd_memory[v], tsfm(mailbox[v]))  (tiger.py:216,352-355).
   * Between the batch that stores a node's message and the batch that consumes it, neither the mailbox row nor
   * the node's memory rows change (both are written only when the node is a positive node of a batch, and that
   * batch consumes the message first), so the reference's on-the-fly h(t'+) of a neighbour is the same row
   * every time it is recomputed.  With this table tg_stream_step computes it ONCE, at the end of the step
   * that stores the message (one updater launch over the unique positive nodes), and STEP 1-2 become a pure
   * gather  reprs[u] = has_msg[v] ? pending[v] : right[v].  Results are those of the lazy form; the updater
   * runs on P <= 2B rows per batch instead of on every involved node with a pending message.
   * Contract (the caller's): whenever state or parameters change by any other route (restart, flush, reset,
   * a training step, loading a snapshot) the table is rebuilt with tg_apply_messages over the has_msg set
   * before the next eager step.  tg_train_step ignores the table. */
  float* pending_vals;
  /* Optional (NULL = identity): PHYSICALLY PARTITIONED state (multi-GPU, www2023tiger_amd/dist.py).  row_of[v] is the row
   * of node v in THIS process's state tables - memories, their times / active flags, mailbox rows and times, the
   * has-message bitmap (bit index = row) and pending_vals - which then have fewer rows than n_nodes: row 0 is the padding
   * node, rows 1..n_own the nodes this rank owns, the rows behind them an arena that holds, for the duration of one batch,
   * the rows pulled from other owners (the caller points row_of at them before the step).  Node ids everywhere else - the
   * T-CSR, the batch arrays, neighbour lists, feature tables, the owner table - stay global.  Honoured by the embedding
   * step (tg_stream_step with embed_only + lean on a model with pending_vals) and by the planned, owner-filtered
   * tg_stream_writeback.  Row-addressed by contract (their id lists are ROWS of this process's tables on such a model -
   * the caller translates its node lists once, when it plans a batch): tg_serve_rows, tg_adopt_rows, tg_gather_eff_rows,
   * tg_apply_messages.  Every other entry point that addresses state by node id - tg_mailbox_consume_gather,
   * tg_consume_update_right(_rows), tg_store_events, tg_restart_seq_fwd(_train), tg_restart_apply, tg_train_step, the full
   * tg_stream_step, tg_attn_gtab_rows - returns TG_EUNSUPPORTED for a model that carries it. */
  const int32_t* row_of;
  /* Optional (NULL = off; needs attn_fused and pending_vals): EAGER QUERY ROWS for streaming with FIXED parameters.
   * g_table [n_nodes, n_head * kvw'] (kvw' as in attn_fused) holds, for every node v, the folded query of the first
   * attention layer  G_v = (e(v) + nfeat(v)) Wqk^T + gconst  with e(v) the node's effective state row (has_msg ?
   * pending : right).  That row - hence G_v - changes only when v is a positive node of a batch (it IS pending[v] from
   * the moment the eager updater writes it, and STEP 4 later copies the same values into the right memory), so a full
   * eager tg_stream_step refreshes the rows of the batch's unique positive nodes at its very end (tg_attn_gtab_rows)
   * and the G product over all 3B centres of the next batches (temporal_agg_modules.py:210-227, the query side) becomes a
   * row lookup by node id.  Same arithmetic per row, so the same bits.  Honoured only by such steps (not with `lazy`,
   * `inner`, embed_only); contract as for pending_vals: rebuilt (all rows) when state or parameters change elsewhere. */
  float* g_table;
  /* Optional, with g_table (NULL = off): CENTRE ROWS.  c_table [n_nodes, d] holds c_v = e(v) + nfeat(v) for every node -
   * what the attention reads of a node both as a centre (the `c` segment of the merger's fc1, tiger.py:196-221 via
   * temporal_agg_modules.py:48-50) and as the node part of a key row (temporal_agg_modules.py:52-66).  It changes exactly
   * when g_table's row does and is kept the same way: the step's eager updater writes the rows of the batch's positive
   * nodes (they are its outputs plus the node features), everything else rebuilds all rows (tg_attn_gtab_rows).  A step
   * that uses the query-row table then gathers ONE row per neighbour for the node part of a key (no has-message test, no
   * pending / right choice, no feature add) and forms no per-batch copy of the centre rows. */
  float* c_table;
} tg_model;

/* Inference-time algebra on the attention weights (parameters only, no data):
 *   g_h   = alpha Wk_h^T (Wq[:, :d] c + qconst_h)            ->  G = c Wqk^T + gconst      (q and g products merged)
 *   fc1([Wo concat_h(Wv_h s_h + bv_h) + bo | c])             ->  [S | c] W1f^T + b1 + valid * c1
 * `fused` receives tg_attn_fused_floats(m) floats: Wqk [n_head*kvw, d], gconst [n_head*kvw],
 * W1f [d, n_head*kvw + d], b1 [d], c1 [d]  (kvw = 2d + d_e).  Must be recomputed whenever an attention
 * parameter or the time encoder changes; training (tg_train_step) ignores it.
 * Where the attention block fits one workgroup (d, d_e <= 256, the tiles of 16 centres inside one CU's 160 KB of LDS)
 * the same weights and fc2 follow once more in FRAGMENT-MAJOR order (csrc/tg_tile.h) and the forward pass of a model
 * carrying `attn_fused` runs the whole block - G product, neighbour gather / softmax, merged value-out-fc1 product,
 * fc2 (temporal_agg_modules.py:48-81,210-235; basic_modules.py:16-19) - as ONE launch per batch with G and S in LDS
 * only (k_attn_tile).  tg_attn_tile_applies: 1 when that form will be taken for `m` (TG_ATTN_TILE=0 switches it off). */
size_t tg_attn_fused_floats(const tg_model* m);
size_t tg_attn_fuse_workspace_bytes(const tg_model* m);
int tg_attn_fuse(const tg_model* m, float* fused, void* ws, size_t ws_bytes, void* stream);
int tg_attn_tile_applies(const tg_model* m);
/* G rows of the listed nodes into m->g_table (see tg_model.g_table): n (<= *n_dev when given) node ids, state rows read as
 * the attention centres read them; ws: n * d floats. */
int tg_attn_gtab_rows(const tg_model* m, int64_t n, const int64_t* nids, const int32_t* n_dev, void* ws, size_t ws_bytes,
                      void* stream);

/* TimeEncode.forward (time_encoding.py:24-26): out[i,:] = cos(fl32(ts[i]*w) + phi) */
int tg_time_encode(int64_t n, const float* ts, int32_t d, const float* freq, const float* phase,
                   float* out, void* stream);

/* Memory.get (memory.py:36-39) / F.embedding: out[i,:] = table[ids[i],:], ts_out optional */
int tg_gather_rows(int64_t n, const int64_t* ids, int32_t width, const float* table, float* out,
                   const float* ts_table, float* ts_out, void* stream);

/* Memory.set (memory.py:41-52): table[ids[i]] = vals[src_index ? src_index[i] : i],
 * ts likewise, active = 1.  n may come from the device (n_dev != NULL, n = capacity).
 * With check != 0 the monotonic-time test sets TG_ERR_PAST_MEMORY in *err. */
int tg_memory_scatter(int64_t n, const int32_t* n_dev, const int64_t* ids, const int64_t* src_index,
                      int32_t width, const float* vals, const float* ts, float* table, float* ts_table,
                      uint8_t* active, int32_t check, uint32_t* err, void* stream);
/* same, with separate row indices for vals (val_index) and ts (ts_index) */
int tg_memory_scatter2(int64_t n, const int32_t* n_dev, const int64_t* ids, const int64_t* val_index,
                       const int64_t* ts_index, int32_t width, const float* vals, const float* ts, float* table,
                       float* ts_table, uint8_t* active, int32_t check, uint32_t* err, void* stream);

/* torch.nn.Linear forward on dense rows: out[n, out_f] = act(x[n, in_f] W^T + b)
 * (message functions message_modules.py:29-55, MergeLayer basic_modules.py:16-19). */
int tg_linear_fwd(int64_t n, const float* x, int32_t in_f, const tg_linear* lin, int32_t out_f, int32_t relu,
                  float* out, void* stream);

/* Its backward (the operator path under autograd): dx [n, in_f] = dy W (NULL: not wanted), dw [out_f, in_f] = dy^T x and
 * db [out_f] = column sums of dy (NULL: not wanted; overwritten, not accumulated).  in_f and out_f multiples of 4. */
size_t tg_linear_bwd_workspace_bytes(int32_t in_f, int32_t out_f);
int tg_linear_bwd(int64_t n, const float* x, int32_t in_f, const float* w, int32_t out_f, const float* dy, float* dx,
                  float* dw, float* db, void* ws, size_t ws_bytes, void* stream);

/* torch.nn.GRUCell forward on dense rows (GRUUpdater.forward, update_modules.py:33-37):
 * x [n, xw] messages, h [n, d] old memory, out [n, d]. */
int tg_gru_fwd(int64_t n, const float* x, int32_t xw, const float* h, int32_t d, const float* w_ih,
               const float* w_hh, const float* b_ih, const float* b_hh, float* out, void* stream);

/* ------------------------------------------------------------------------- */
/* STEP 1-2: consume pending messages (tiger.py:208-221,292-356)              */
/* ------------------------------------------------------------------------- */
/* reprs[u,:] = right_vals[involved[u],:] for u < *n_involved
 * (tiger.py:214, LastMessageAggregatorNoGradLastOnly gather message_modules.py:155-156) */
int tg_mailbox_consume_gather(const tg_model* m, const int64_t* involved, const int32_t* n_involved,
                              int64_t cap, float* reprs, void* stream);

/* h(t'+) for the outdated nodes: msgs = tsfm(mailbox[ids]); h = updater(upd_mem[ids], msgs);
 * reprs[out_pos[o],:] = h.  Also checks the mailbox/memory time invariants.
 * (tiger.py:319-336,352-355; update_modules.py:30-47; message_modules.py:20-55) */
size_t tg_apply_messages_workspace_bytes(const tg_model* m, int64_t cap);
int tg_apply_messages(const tg_model* m, const int64_t* outdated, const int32_t* out_pos,
                      const int32_t* n_outdated, int64_t cap, float* reprs, uint32_t* err, void* ws,
                      size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* STEP 3: temporal-attention embedding, one layer                            */
/* (temporal_agg_modules.py:29-83,186-235; basic_modules.py:16-19)            */
/* ------------------------------------------------------------------------- */
/* reprs [U,d] are the involved nodes' h(t'+); (bitmap, rank) give the local index.
 * centre ids nids[Q], query times ts[Q] (float32), neighbours l1_* [Q,K].
 * out [Q,d]. */
size_t tg_temporal_attn_workspace_bytes(const tg_model* m, int64_t Q);
int tg_temporal_attn_fwd(const tg_model* m, int64_t Q, const int64_t* nids, const float* ts,
                         const int64_t* l1_nids, const int64_t* l1_eids, const float* l1_ts,
                         const float* reprs, const uint64_t* bitmap, const uint32_t* rank, float* out,
                         void* ws, size_t ws_bytes, void* stream);

/* The FIRST attention layer of --n_layers 2 (temporal_agg_modules.py:29-83 at depth == n_layers; weights fns[0]):
 * the node part of key k of centre i is key_rows[i*K + k, :] - the embedding of that neighbour computed by the layer
 * below (tg_temporal_attn_fwd on the Q*K neighbours as centres with the weights of fns[1], hop-2 neighbours sampled at
 * the neighbours' timestamps, data_loader.py:131, query time = the root's, :63) - instead of its memory row + node
 * features; edge features, time encoding and the padding mask (l1_nids == 0) are as in tg_temporal_attn_fwd. */
int tg_temporal_attn_fwd_keys(const tg_model* m, int64_t Q, const int64_t* nids, const float* ts,
                              const int64_t* l1_nids, const int64_t* l1_eids, const float* l1_ts,
                              const float* reprs, const uint64_t* bitmap, const uint32_t* rank,
                              const float* key_rows, float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* STEP 4-6: write-back (tiger.py:229-255,396-442; memory.py:77-106)          */
/* ------------------------------------------------------------------------- */
/* STEP 4: for every unique positive node that is outdated: right memory <- its
 * h(t'+) row of reprs, update_ts <- mailbox ts, has-message bit cleared. */
int tg_consume_update_right(const tg_model* m, const int64_t* upos, const int32_t* n_upos, int64_t cap,
                            const float* reprs, const uint64_t* bitmap, const uint32_t* rank,
                            uint32_t* err, void* stream);
/* same, reading node upos[p]'s new row at rows[row_index[p]] instead of reprs[local(upos[p])] */
int tg_consume_update_right_rows(const tg_model* m, const int64_t* upos, const int32_t* n_upos, int64_t cap,
                                 const float* rows, const int64_t* row_index, uint32_t* err, void* stream);
/* Effective right-memory rows: out[i] = has_msg[v] ? pending_vals[v] : right_vals[v], ts_out[i] = has_msg[v] ?
 * msg_ts[v] : right_ts[v] for v = ids[i] - the row and time STEP 4 would leave in the right memory if v's pending
 * message were consumed now (tiger.py:214-221,236-241).  Needs tg_model.pending_vals (eager updates).  This is
 * what the owner of a node serves to a rank that embeds an event involving it (partitioned multi-GPU path). */
int tg_gather_eff_rows(const tg_model* m, int64_t n, const int64_t* ids, float* out, float* ts_out, void* stream);

/* STEP 5: build the two raw messages of the winning event of each unique positive
 * node and write mailbox row, mailbox ts and has-message bit.  `index` is the
 * select_latest position into cat[src,dst]. */
int tg_store_events(const tg_model* m, int64_t B, const int64_t* src, const int64_t* dst, const float* ts,
                    const int64_t* eids, const int64_t* upos, const int64_t* index, const int32_t* n_upos,
                    uint32_t* err, void* stream);

/* ------------------------------------------------------------------------- */
/* Restarters (restarters.py:36-114,254-277) and TIGER.restart (tiger.py:594-609) */
/* ------------------------------------------------------------------------- */
typedef struct tg_seq_restarter {
  int32_t hist_len;
  int32_t n_head;
  const float* te_freq;  /* the restarter's own TimeEncode (restarters.py:27) */
  const float* te_phase;
  const float* anony_emb; /* [hist_len+1, d] */
  const float* in_proj_w; /* [3*dm, dm], dm = 3d + d_e + d */
  const float* in_proj_b;
  tg_linear out_proj;     /* [dm, dm] */
  tg_linear out_fn;       /* [d, dm]  */
  tg_linear fc1, fc2;     /* merger: [d, d + dm - d], [d, d] */
  /* != 0: the caller knows that tg_model.nfeats is all zeros (every JODIE data set: feature_getter.py:25-47 loads a zero
   * table).  The Q / K projection then skips the two node-feature column blocks and tabulates the anony_emb block
   * (K = d_e + d instead of dm; csrc/tg_restart.hip).  0 is always correct. */
  int32_t nfeats_zero;
  int32_t reserved;
  /* Optional (NULL = computed per call), with nfeats_zero: T_a = anony_emb in_proj_w[0:2dm, 2d:3d]^T ([hist_len + 1, 2 dm], no
   * bias) precomputed by the caller for FIXED parameters (inference): one product less per restart.  The caller recomputes
   * it whenever anony_emb or in_proj_w change; the training entry points ignore it. */
  const float* ta_cached;
} tg_seq_restarter;

size_t tg_restart_seq_workspace_bytes(const tg_model* m, const tg_seq_restarter* r, int64_t n);
/* SeqRestarter.forward given the collated history (hist_* [n,H], anonymized ids):
 * h_left, h_right [n,d], prev_ts [n]. */
int tg_restart_seq_fwd(const tg_model* m, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                       const int64_t* hist_nids, const int64_t* anon_ids, const int64_t* hist_eids,
                       const float* hist_ts, const int64_t* hist_dirs, float* h_left, float* h_right,
                       float* prev_ts, void* ws, size_t ws_bytes, void* stream);

/* The same forward in training mode (the reference calls TIGER.restart inside the training loop with
 * the module in train() mode, so the restarter's attention / merger dropout is active): masks as
 * described at tg_train_io; rng[1] is incremented.  dropout_p == 0 equals tg_restart_seq_fwd. */
int tg_restart_seq_fwd_train(const tg_model* m, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                             const int64_t* hist_nids, const int64_t* anon_ids, const int64_t* hist_eids,
                             const float* hist_ts, const int64_t* hist_dirs, float* h_left, float* h_right,
                             float* prev_ts, float dropout_p, uint64_t* rng, void* ws, size_t ws_bytes,
                             void* stream);

/* TIGER.restart with the SeqRestarter as ONE call over a device-resident node list (tiger.py:594-609 + restarters.py:51-114):
 * histories of the n nodes at time *t_dev (float32, the batch's earliest time: eval_utils.py:37-42 restarts every node at
 * ts.min()) sampled with the recent-edges strategy, anonymised ids, the restarter's forward (inference form) and
 * tg_restart_apply.  Same kernels as the calls it replaces; for loops that restart per batch (the lazy restart of the
 * evaluation harness), where a dozen library calls per batch were the cost. */
size_t tg_restart_seq_list_workspace_bytes(const tg_model* m, const tg_seq_restarter* r, int64_t n);
int tg_restart_seq_list(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                        const float* t_dev, void* ws, size_t ws_bytes, void* stream);
/* The same in train() mode (the reference calls TIGER.restart inside the training loop with the module in train() mode:
 * attention / merger dropout active, masks as tg_restart_seq_fwd_train draws them; rng[1] is incremented). */
int tg_restart_seq_list_train(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                              const float* t_dev, float dropout_p, uint64_t* rng, void* ws, size_t ws_bytes, void* stream);
/* The same with the live count on the device: the launches are sized for `cap` entries, the first *n_dev (<= cap) are
 * restarted; the entries behind them must hold valid node ids.  No host value depends on the count, so the call can be
 * captured into a hipGraph and replayed for every batch whose count fits the capacity. */
int tg_restart_seq_list_dev(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t cap, const int64_t* nids,
                            const int32_t* n_dev, const float* t_dev, void* ws, size_t ws_bytes, void* stream);
/* The forward alone: h_left, h_right [n,d] and prev_ts [n] of the listed nodes into the caller's buffers, NO state update
 * (tg_restart_apply is the caller's, on whichever stream the state lives).  It reads the graph, the feature tables and the
 * restarter's parameters only - nothing a streaming step writes - so a loop that restarts per batch can run it on a second
 * stream beside the previous batch's step (eval_utils._RestartPipeline).  n_dev as above (NULL: all n).  Same workspace. */
int tg_restart_seq_list_fwd(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int64_t n, const int64_t* nids,
                            const int32_t* n_dev, const float* t_dev, float* h_left, float* h_right, float* prev_ts,
                            void* ws, size_t ws_bytes, void* stream);
/* The same forward over SEVERAL lists at once, each with its own time (the lists of consecutive batches: their restarts are
 * independent of each other and of the steps between them - a node listed for batch k + 1 is not involved in batch k - and
 * one forward over all of them costs little more than one over the shortest).  The ids go to ids_out concatenated in list
 * order (empty lists skipped), rows i of h_left / h_right / prev_ts belong to ids_out[i]; workspace for sum(counts) nodes. */
#define TG_RESTART_MAX_LISTS 8
int tg_restart_seq_lists_fwd(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, int32_t n_lists,
                             const int64_t* const* lists, const int64_t* counts, const float* const* t_dev,
                             int64_t* ids_out, float* h_left, float* h_right, float* prev_ts, void* ws, size_t ws_bytes,
                             void* stream);
/* The StaticRestarter (restarters.py:254-277) in the same form: h_left / h_right = the rows of its two tables, prev_ts = the
 * time of the node's last event strictly before the list's time (0: none).  One launch, no workspace. */
int tg_restart_static_lists_fwd(const tg_model* m, const tg_tcsr* g, const float* static_left, const float* static_right,
                                int32_t n_lists, const int64_t* const* lists, const int64_t* counts,
                                const float* const* t_dev, int64_t* ids_out, float* h_left, float* h_right, float* prev_ts,
                                void* stream);

/* TIGER.restart's state update (tiger.py:603,608-609): clear has-message bits of
 * nids, then left/right memory rows and timestamps <- (h_left, h_right, prev_ts)
 * with skip_check semantics. */
int tg_restart_apply(const tg_model* m, int64_t n, const int64_t* nids, const float* h_left,
                     const float* h_right, const float* prev_ts, void* stream);

/* ------------------------------------------------------------------------- */
/* Fused streaming step: collate + STEP 1-6 of TIGE.contrast_learning         */
/* (data_loader.py:77-131 + tiger.py:196-255) with no host round trip.         */
/* ------------------------------------------------------------------------- */
struct tg_lazy_restart; /* below */
typedef struct tg_step_io {
  int64_t B;
  const int64_t* src;   /* [B] */
  const int64_t* dst;   /* [B] */
  const int64_t* neg;   /* [B] */
  const double* ts;     /* [B] float64 event times (sampler side) */
  const int64_t* eids;  /* [B] */
  /* outputs (all optional except h) */
  float* h;             /* [3B, d] temporal embeddings of cat[src,dst,neg]; rows [0,2B) = h_left */
  int64_t* l1_nids;     /* [3B, K] */
  int64_t* l1_eids;     /* [3B, K] */
  float* l1_ts;         /* [3B, K] */
  int64_t* involved;    /* [3B*(K+1)] capacity */
  int32_t* counts;      /* [4]: n_involved, n_outdated, n_unique_pos, n_restarted (lazy restart, else 0) */
  float* h_prev_left;   /* [2B, d] restarter targets (tiger.py:248-251) or NULL */
  float* h_prev_right;  /* [2B, d] or NULL */
  uint32_t* err;        /* invariant word */
  /* Resident-stream mode: when offset_dev != NULL, src/dst/neg/ts/eids point at the
   * whole time-ordered stream in HBM and the batch is elements [*offset_dev, *offset_dev+B);
   * with advance != 0 the step ends by adding B to *offset_dev, so a captured hipGraph of
   * one step replays the whole stream with no host work (train_self_supervised.py:143). */
  int64_t* offset_dev;
  int32_t advance;
  /* embed_only != 0: stop after STEP 3 (no state is written).  Used by the multi-GPU
   * path, where every rank embeds its own shard of the batch and the write-back runs on
   * the all-gathered rows (www2023tiger_amd/dist.py). */
  int32_t embed_only;
  void* profiler;       /* tg_profiler* or NULL: records an event after every stage */
  float* h_new;         /* [2B, d] or NULL: h(t'+) of cat[src,dst] (the rows STEP 4 would write) */
  /* ws_is_clean != 0: the caller guarantees that the first tg_stream_step_zero_bytes() bytes of
   * the workspace are zero (freshly zero-filled, or left by a previous completed full step, which
   * always cleans up after itself); the step then skips its initial memset launch. */
  int32_t ws_is_clean;
  /* rows_hint > 0: the caller's bound on the number of nodes with a pending message among the involved nodes of
   * this batch (e.g. 1.5 x the largest count seen so far).  Performance only: it lets the updater pick blocks
   * sized for a launch that fits the chip in one round; a batch that exceeds the bound is still correct.
   * 0 = unknown (the capacity and the node count are used).  With eager updates (tg_model.pending_vals) the updater
   * runs on the unique positive nodes of the batch and the bound is on THEIR number (counts[2]).  (This field was
   * `reserved` before: 0 is the old behaviour, the layout is unchanged.) */
  int32_t rows_hint;
  /* Lazy restart of train_self_supervised.py:152-163 with the StaticRestarter, on device (NULL = off);
   * see tg_lazy_restart below. */
  const struct tg_lazy_restart* lazy;
  /* collate_only != 0: stop after the collation (sampler + involved-set compaction): l1_* / involved / counts are
   * the outputs, no state is read or written, h may be NULL.  The partitioned multi-GPU path uses it to learn
   * which rows a rank must pull from their owners before it embeds (www2023tiger_amd/dist.py). */
  int32_t collate_only;
  /* eager_copy != 0 (only meaningful with tg_model.pending_vals): keep the compact copy of the involved rows - STEP 1-2
   * as ONE stand-alone gather launch into reprs - instead of letting the attention launches read pending / right by
   * node id (the default "direct" form, one launch and one row copy fewer).  Same results. */
  int32_t eager_copy;
  /* lean != 0: the caller does not need the involved / outdated SETS of the batch (io->involved is ignored, counts[0] and
   * counts[1] come back as -1).  With eager updates in the direct form nothing else in the step needs them either - rows
   * are addressed by node id, the dedup slots can be indexed by node id, the time invariants can be checked per centre /
   * per neighbour - so the sampler marks no flags and the compaction launch is skipped.  Honoured only there (and only
   * without the lazy-restart loop and the h_prev_* outputs); ignored otherwise.  Same results.  An embed_only step
   * honours it too (then `counts` is not written at all and the stream offset is advanced by the core launch). */
  int32_t lean;
  /* Graph.sample_temporal_neighbor's strategy for the neighbours of the batch (graph.py:94-148; init_utils.py:40):
   * 0 = recent_edges (the default recipe), 1 = recent_nodes (last occurrence of each distinct neighbour, graph.py:129-143).
   * 2 = uniform (graph.py:101-115): K draws of numpy's legacy randint per non-empty query, consumed from the graph's
   * MT19937 stream (`mt_state` below) in query order, sorted by time.  With `inner` the second hop follows the same strategy
   * (uniform: the stream goes on behind the first hop's draws).  2 not together with `lazy` (a collate-only pass would consume
   * draws the step repeats). */
  int32_t strategy;
  /* --n_layers 2 (tiger.py:29; data_loader.py:105-131; temporal_agg_modules.py:29-83).  NULL: one attention layer.
   * Otherwise a tg_model that differs from the step's model only in its attention block: the weights of the SECOND
   * layer (temporal_embedding_fn.fns[1]; attn_fused optional, as for the first).  The step then samples the second hop
   * for every neighbour slot at the neighbour's own float32 timestamp (data_loader.py:131), counts those nodes among
   * the involved ones, embeds the Q*K neighbour slots with that layer at the ROOT's query time
   * (temporal_agg_modules.py:57-66) and feeds their embeddings to the first layer as the node part of its keys.
   * Workspace: tg_stream_step_workspace_bytes2(m, B, 2). */
  const struct tg_model* inner;
  /* Collate prefetch (resident-stream mode; both fields 0 / NULL = off).  The collate part of a batch - temporal
   * neighbour sampling (data_loader.py:77-131, graph.py:94-127), the centre rows, the first dedup pass and the pre-batch
   * snapshot - reads the graph and state that is final once the previous batch's updater has run.  With prefetch_state
   * set, a full lean eager step of a model with eager query rows runs that part for the NEXT batch (the events at the
   * advanced offset) as extra workgroups of its own last launch, and the next call starts with its attention core: one
   * launch per batch less, same results.  The neighbour lists then live in the workspace only: l1_nids / l1_eids / l1_ts
   * must be NULL (as outputs they would hold the NEXT batch's lists when the call returns).
   *   stream_len:      number of events in the resident stream arrays (a batch past the end is not prefetched: the
   *                    rider does nothing and the batch must not be run);
   *   *prefetch_state: HOST int, in / out.  In: 1 - the previous call on this workspace prefetched this batch and
   *                    neither state, graph, offset nor workspace were touched since (the caller's promise); 2 - it
   *                    prefetched, but that work must be discarded (the step then clears the dedup slots the prefetch
   *                    marked and collates itself; a step that cannot use a prefetch - not lean, no eager query rows -
   *                    treats 1 the same way); 0 - nothing was prefetched.  Out: 1 - this call prefetched the next
   *                    batch, else 0.  A captured hipGraph replays whatever the capturing call did: capture with
   *                    *prefetch_state == 1 on entry and exit. */
  int64_t stream_len;
  int32_t* prefetch_state;
  /* Debug outputs (ABI 7; NULL = off): the neighbour lists of the batch THIS call consumes - [3B, K] each, as l1_nids /
   * l1_eids / l1_ts - copied out of the workspace before the step's last launch replaces them with the next batch's.
   * They let the collate-prefetch form, whose l1_* outputs must be NULL, be checked for bit-exact neighbour indices
   * (graph.py:67-148) exactly as it is timed. */
  int64_t* dbg_l1_nids;
  int64_t* dbg_l1_eids;
  float* dbg_l1_ts;
  /* strategy == 2 (`uniform`, graph.py:101-115): the graph's numpy RandomState on the device - uint32 [625]: the 624 key
   * words and the position, as tg_sample_uniform takes it; the step draws from it in query order (cat[src, dst, neg]) and
   * leaves it where the reference's generator would stand after the batch. */
  uint32_t* mt_state;
} tg_step_io;

/* The reference loop draws `np.random.rand() < restart_prob` before every batch but the first; a hit sets
 * `restarting`, forgets which nodes are up to date and drops every pending message (msg_store.clear()).  While
 * restarting, every batch re-initialises its involved nodes that are not yet up to date with
 * TIGER.restart(nodes, full(min(ts))) (tiger.py:594-609).  With the StaticRestarter (restarters.py:254-277) the
 * surrogate state is two table rows and the time of the node's last event before min(ts), so the whole
 * loop body runs inside tg_stream_step, between the sampler and STEP 1, without a host round trip:
 *   trigger[*batch_dev]  != 0: uptodate bitmap and has-message bitmap cleared, *restarting_dev = 1;
 *   *restarting_dev != 0: for every involved node v without its uptodate bit:
 *       left/right memory rows <- static_left[v] / static_right[v], both update_ts <- t'(v) = time of v's last
 *       event strictly before float32(min(ts of the batch)) (0 when there is none), has-message bit cleared,
 *       uptodate bit set.
 * The step ends by incrementing *batch_dev.  The draws are the caller's (pre-drawn per batch, so a run is
 * reproducible and can be replayed as a hipGraph). */
typedef struct tg_lazy_restart {
  const float* static_left;   /* StaticRestarter.left_emb.weight  [n_nodes, d] */
  const float* static_right;  /* StaticRestarter.right_emb.weight [n_nodes, d] */
  const uint8_t* trigger;     /* [n_trigger]; batches at or beyond n_trigger do not trigger */
  int64_t n_trigger;
  int64_t* batch_dev;         /* device batch counter (index into trigger); NULL = index 0, not advanced */
  int32_t* restarting_dev;    /* device flag, persists between steps */
  uint64_t* uptodate;         /* device bitmap over n_nodes (tg_bitmap_words), persists between steps */
  /* LIST form (static_left == static_right == NULL; any restarter, e.g. the SeqRestarter of the reference's default
   * recipe, restarters.py:36-114): the loop's bookkeeping runs on the device - trigger, bitmaps, has-message bits as
   * above - but instead of writing surrogate rows the step stores the ids of the nodes to re-initialise in
   * list[0 .. counts[3]) (order unspecified) and float32(min(ts of the batch)) in *tmin.  Used with
   * tg_step_io.collate_only: the caller reads counts[3] (4 bytes), runs its restarter on the list
   * (tg_restart_seq_fwd + tg_restart_apply) and then the step itself. */
  int64_t* list;              /* [>= min(3B(K+1), n_nodes)] */
  float* tmin;                /* [1] */
  /* List form only: != 0 leaves the has-message bitmap alone while no trigger fires - the caller's tg_restart_apply clears
   * the bits of the listed nodes anyway.  The pass then touches nothing a streaming step reads or writes (graph, batch
   * arrays, the up-to-date bitmap, its own outputs) and may run on another stream beside one.  A firing trigger still
   * clears the whole bitmap: a caller that overlaps passes and steps must not pre-draw triggers. */
  int32_t keep_msg_bits;
  int32_t reserved;
} tg_lazy_restart;

/* Per-stage timer of tg_stream_step (HIP events on the step's stream).  Stage names:
 * tg_profiler_stage_name(i), i < tg_profiler_num_stages().  tg_profiler_read waits for
 * the last event and returns the elapsed milliseconds of every stage of the last step
 * that ran with this profiler attached.  Not capturable into a hipGraph. */
typedef struct tg_profiler tg_profiler;
tg_profiler* tg_profiler_create(void);
void tg_profiler_destroy(tg_profiler* p);
int tg_profiler_num_stages(void);
const char* tg_profiler_stage_name(int stage);
int tg_profiler_read(tg_profiler* p, float* ms_out);
/* Kernel-bound durations of the same step: while the profiler is attached the step's main kernels are launched with an
 * event pair bound to the dispatch (its own begin / end timestamps, what rocprofv3 reports), one slot per kernel:
 * tg_profiler_kernel_slot_name(i), i < tg_profiler_num_kernel_slots().  ms_out[slot] < 0: the step made no launch under
 * the slot; names_out (nullable) receives the launch expression of the timed kernel (template arguments included). */
int tg_profiler_num_kernel_slots(void);
const char* tg_profiler_kernel_slot_name(int slot);
int tg_profiler_kernel_ms(tg_profiler* p, float* ms_out, const char** names_out);

size_t tg_stream_step_workspace_bytes(const tg_model* m, int64_t B);
size_t tg_stream_step_workspace_bytes2(const tg_model* m, int64_t B, int32_t n_layers); /* n_layers 1 or 2 */
/* size of the leading workspace region that must be zero when a step starts (a caller that sets ws_is_clean clears
 * exactly this prefix; with io->inner - two layers - the region is larger: use the second form) */
size_t tg_stream_step_zero_bytes(const tg_model* m, int64_t B);
size_t tg_stream_step_zero_bytes2(const tg_model* m, int64_t B, int32_t n_layers); /* n_layers 1 or 2 */
int tg_stream_step(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, void* ws, size_t ws_bytes,
                   void* stream);
/* The form tg_stream_step(m, ., io, ..) takes, as a mask of TG_FORM_*: the library's own decision (field values AND its
 * tuning knobs), so that a host that keeps derived tables does not have to restate it.  TG_FORM_TABLES: the step reads
 * the per-node tables (tg_model.g_table / c_table) and leaves them current - rows of the batch's positive nodes and, with
 * the in-step restart loop, of the nodes it re-initialises; a step WITHOUT the bit on a model that carries the tables
 * leaves them stale wherever it changes state (the in-step restart loop, tg_lazy_restart), and the host must rebuild
 * them (tg_attn_gtab_rows) before a later step that has the bit. */
#define TG_FORM_DIRECT 1   /* rows read from pending / right by node id: no compact copy of the involved rows */
#define TG_FORM_FUSED_WB 2 /* STEP 4-6 as one pass (possibly riding on the attention block's launches) */
#define TG_FORM_LEAN 4     /* no involved / outdated sets are formed */
#define TG_FORM_TABLES 8   /* reads and maintains g_table / c_table */
int32_t tg_stream_step_form(const tg_model* m, const tg_step_io* io);

/* ------------------------------------------------------------------------- */
/* Training tail: STEP 7, backward pass, Adam            */
/* tiger.py:257-288 (scores + BCE), tiger.py:547-592 (mutual loss),            */
/* train_self_supervised.py:165-171 (loss.backward(); optimizer.step())        */
/* ------------------------------------------------------------------------- */
#define TG_HIT_NONE 0
#define TG_HIT_VEC 1
#define TG_HIT_BIN 2
#define TG_HIT_COUNT 3

/* score head of TIGE (tiger.py:135-149): optional hit embedding + MergeLayer(W, W, d, 1),
 * W = d (+ n_neighbors for 'vec') */
typedef struct tg_score_params {
  int32_t hit_type;   /* TG_HIT_* */
  int32_t n_hit_rows; /* rows of hit_emb: 2 ('bin'), n_neighbors + 1 ('count'), else 0 */
  const float* hit_emb; /* [n_hit_rows, d] or NULL */
  tg_linear fc1;      /* [d, 2W] */
  tg_linear fc2;      /* [1, d]  */
} tg_score_params;

/* One training iteration's device work: the fused step (as tg_stream_step) with STEP 7 and the
 * backward pass of the contrastive loss inserted between STEP 3 and the write-back.
 * Gradients are ACCUMULATED (+=) into buffers laid out like the parameters: `grads` is a
 * tg_model whose parameter pointers (te_*, gru_*, attn_*) point at gradient buffers (other
 * fields ignored), `score_grads` likewise for the score head.  Supported: message transform
 * 'id', updater 'gru', one attention layer.
 * grads == NULL selects EVALUATION: forward, STEP 7 scores and BCE loss, write-back - no gradients,
 * no mutual loss, no dropout (the path of tiger/eval_utils.py:29-48); attn_fused is honoured.
 * flags (device int32[4]) says which parameter groups received a gradient this step (torch leaves
 * the .grad of the others None and Adam skips them): [0] = 1 always (embedding, score head, time
 * encoder), [1] = the GRU ran (some involved node had a pending message), [2] = the mutual loss
 * had at least one valid target row (restarter parameters), [3] reserved. */
typedef struct tg_train_io {
  tg_step_io step;
  const tg_score_params* score;
  const tg_model* grads;
  const tg_score_params* score_grads;
  float* losses;     /* [2] device: contrast loss (mean BCE over 2B logits), mutual loss */
  float* pos_scores; /* [B] logits or NULL */
  float* neg_scores; /* [B] or NULL */
  int32_t* flags;    /* [4] device, see above (NULL allowed) */
  /* mutual learning (tiger.py:574-590): the restarter predicts the targets h_prev_left/right
   * (step.h_prev_* must be given) of the latest occurrence of every positive node; MSE over the
   * rows whose target is not all zero.  Restart data (data_loader.py:133-165) is collated on
   * device from the graph passed to tg_train_step - the collator's graph, as in the reference,
   * where the histories of the mutual loss come from the collated batch (tiger.py:579-581).  restarter = TG_RESTARTER_NONE is contrast_only (tiger.py:570-572). */
  int32_t restarter; /* TG_RESTARTER_* */
  int32_t reserved;
  const tg_seq_restarter* seq;        /* TG_RESTARTER_SEQ: parameters ... */
  const tg_seq_restarter* seq_grads;  /* ... and gradient buffers in the same layout (+=) */
  const float* static_left;           /* TG_RESTARTER_STATIC: left_emb / right_emb tables [n_nodes, d] */
  const float* static_right;
  float* static_left_grad;            /* dense gradients [n_nodes, d] (+=) */
  float* static_right_grad;
  /* dropout (the reference's --dropout: attention probabilities of the embedding and of the
   * SeqRestarter, hidden layer of the score head and of the SeqRestarter's merger).  Masks are a
   * counter-based hash of (rng[0] = seed, rng[1] = step counter, stream, element); the step ends by
   * incrementing rng[1].  The masks are NOT torch's: results match the reference in distribution. */
  float dropout_p;                    /* 0 <= p < 1; 0 = off */
  int32_t reserved2;
  uint64_t* rng;                      /* device uint64[2]; required when dropout_p > 0 */
  /* --n_layers 2 (step.inner != NULL; tiger.py:29, temporal_agg_modules.py:29-83): gradient buffers of the SECOND
   * attention layer's parameters (temporal_embedding_fn.fns[1]) in the layout of `grads`' attention block (+=); the
   * other fields of this tg_model are ignored.  Required for training with two layers. */
  const tg_model* inner_grads;
} tg_train_io;

#define TG_RESTARTER_NONE 0
#define TG_RESTARTER_SEQ 1
#define TG_RESTARTER_STATIC 2

/* seq: the SeqRestarter when restarter == TG_RESTARTER_SEQ (sizes only), else NULL */
size_t tg_train_step_workspace_bytes(const tg_model* m, const tg_score_params* sp, int32_t restarter,
                                     const tg_seq_restarter* seq, int64_t B);
size_t tg_train_step_workspace_bytes2(const tg_model* m, const tg_score_params* sp, int32_t restarter,
                                      const tg_seq_restarter* seq, int64_t B, int32_t n_layers); /* n_layers 1 or 2 */
int tg_train_step(const tg_model* m, const tg_tcsr* g, const tg_train_io* io, void* ws, size_t ws_bytes,
                  void* stream);

/* The evaluation pass in RESTART MODE (eval_utils.py:37-42 inside the loop of :60-75: before every batch, the involved
 * nodes that are not up to date are re-initialised by TIGER.restart at the batch's earliest time) over `count` consecutive
 * batches of a device-resident stream as ONE call, SeqRestarter in inference form (or, r == NULL, the StaticRestarter whose
 * tables the run carries).  Per batch k the host-side loop makes
 * these calls: a collate-only pass in the list form (tg_lazy_restart) lists the nodes, the restarter's forward computes their
 * rows, tg_restart_apply writes them, tg_attn_gtab_rows refreshes their query / centre rows (a model that streams with
 * current per-node tables), tg_train_step in its evaluation form scores the batch.  Here the same calls run on two streams
 * and in GROUPS of `group` batches:
 *   - a pass reads the graph, the batch arrays and the up-to-date bitmap (keep_msg_bits: not the has-message bits), the
 *     restarter's forward the graph, the feature tables and its own parameters - nothing a step writes - so the passes of
 *     group q + 1 and the forward of group q run on a stream of the library's own beside the steps of group q - 1;
 *   - the lists of a group's batches are disjoint (a pass marks what it lists) and a node listed for batch k + 1 is not
 *     involved in batch k (it would have been listed there), so step k neither reads nor writes it: ONE forward over the
 *     group's lists (tg_restart_seq_lists_fwd, every list at its own time) and ONE apply / table refresh ahead of the
 *     group's first step give every step the state the per-batch order gives it - same lists, same marks; the rows bit for
 *     bit at group 1, to rounding beyond (a forward over more rows takes other blocks for its products);
 *   - `stream` keeps the state (apply, table rows, steps); events order the two streams; two halves of 2 * group pass
 *     contexts and two row sets alternate, so that in the steady state neither stream waits for the other's bookkeeping.
 * The host reads one count per batch (pinned memory) to size the forward: not capturable.  Triggers must not fire
 * (keep_msg_bits).  On return `stream` is ordered behind everything enqueued.  The side stream and its events belong to
 * the library, one set per device: one run at a time per device (calls from several host threads must be serialised). */
#define TG_RUN_CTX (2 * TG_RESTART_MAX_LISTS)
typedef struct tg_restart_run {
  int32_t group;                /* batches per forward, 1 .. TG_RESTART_MAX_LISTS; 2 * group pass contexts are used */
  int32_t reserved;
  const tg_step_io* pass_io[TG_RUN_CTX]; /* collate_only + lazy (list form, keep_msg_bits != 0); offset_dev is set per pass */
  void* pass_ws[TG_RUN_CTX];
  size_t pass_ws_bytes[TG_RUN_CTX];
  const tg_tcsr* g_restart;     /* the restarter's graph (histories); the steps and passes sample from `g` */
  const int64_t* offsets;       /* device [count]: stream offset of batch k */
  int64_t* batch_dev;           /* device batch counter of the lazy-restart loop, incremented per pass (NULL: none) */
  int32_t* count_host[TG_RUN_CTX]; /* pinned host memory: the count of the context's last pass */
  int64_t cap;                  /* capacity of each context's list */
  int64_t rows_cap;             /* capacity of a row set: min(group * cap, n_nodes) - the lists of a group are disjoint */
  int64_t* ids[2];              /* [rows_cap] per set: the group's lists, concatenated */
  float* h_left[2];             /* [rows_cap, d] per set: the restarter's rows */
  float* h_right[2];
  float* prev_ts[2];            /* [rows_cap] */
  int64_t fwd_nodes;            /* nodes per forward (> 0; a group with more takes several, each over <= fwd_nodes of them) */
  void* fwd_ws;                 /* tg_restart_seq_list_workspace_bytes(m, r, fwd_nodes); unused with the static restarter */
  size_t fwd_ws_bytes;
  const float* static_left;     /* r == NULL: the StaticRestarter's tables [n_nodes, d] (tg_restart_static_lists_fwd) */
  const float* static_right;
  void* gtab_ws;                /* rows_cap * d floats + 64 bytes, or NULL: no per-node tables to follow */
  size_t gtab_ws_bytes;
  float* pos_scores;            /* [count * B]: step k writes its logits at k * B (NULL: where step_io points) */
  float* neg_scores;
  int32_t* n_restarted;         /* host [count] out (NULL: not wanted): nodes re-initialised before batch k */
  /* collate prefetch inside a group (0: off): the number of events in the resident stream arrays and the stream offset of
   * batch 0 of the run (= offsets[0]) - the steps of a group but its last then prefetch the next batch's collate part as the
   * plain resident pass does (tg_step_io.prefetch_state; the run owns the flag, and the steps' l1_* outputs are dropped) */
  int64_t stream_len;
  int64_t first_offset;
} tg_restart_run;
int tg_eval_restart_run(const tg_model* m, const tg_tcsr* g, const tg_seq_restarter* r, const tg_train_io* step_io,
                        void* step_ws, size_t step_ws_bytes, const tg_restart_run* run, int64_t count, void* stream);

/* torch.optim.Adam (defaults: no weight decay, no amsgrad) over a device-resident table of
 * parameter segments.  Segment i belongs to group `group`; a group whose enabled flag
 * (device int32, NULL = always on) is 0 is skipped entirely, including its step count
 * (steps[group], device int32, incremented here), exactly as Adam skips parameters whose
 * .grad is None.  g is read scaled by `grad_scale`. */
typedef struct tg_adam_seg {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  int32_t group;
  float grad_scale; /* per-segment factor on g (e.g. the mutual-loss coefficient); 0 means 1 */
} tg_adam_seg;

int tg_adam_step(const tg_adam_seg* segs_dev, int32_t n_segs, int32_t n_groups, const int32_t* enabled_dev,
                 int32_t* steps_dev, float lr, float beta1, float beta2, float eps, float grad_scale,
                 void* stream);

/* ------------------------------------------------------------------------- */
/* Evaluation metrics (tiger/eval_utils.py:49-67)           */
/* ------------------------------------------------------------------------- */
/* For consecutive windows of `chunk` events (the last may be shorter): sklearn's
 * average_precision_score and roc_auc_score of the predictions [pos | neg] with labels [1 | 0],
 * ties handled as sklearn does.  pos_pred / neg_pred: [n] probabilities (sigmoid of the logits);
 * ap / auc: double[ceil(n / chunk)] on device.  Non-finite predictions are dropped from their
 * window and counted in *n_nonfinite (nullable, device int32, not reset here). */
int tg_ap_auc(int64_t n, int32_t chunk, const float* pos_pred, const float* neg_pred, double* ap, double* auc,
              int32_t* n_nonfinite, void* stream);

/* ------------------------------------------------------------------------- */
/* Multi-GPU: replicated write-back of a GLOBAL batch from all-gathered rows  */
/* (www2023tiger_amd/dist.py; STEP 4-6 of tiger.py:229-255 for every event of */
/* the global batch, the embeddings having been computed on other ranks)      */
/* ------------------------------------------------------------------------- */
typedef struct tg_writeback_io {
  int64_t Bg;               /* events of the global batch */
  const int64_t* src;       /* resident global stream (or the batch itself when offset_dev == NULL) */
  const int64_t* dst;
  const double* ts;
  const int64_t* eids;
  int64_t* offset_dev;      /* element offset of this batch in the stream arrays; += Bg when advance != 0 */
  int32_t advance;
  int32_t reserved;
  const float* rows;        /* gathered rows [*, d] */
  const int64_t* left_row;  /* [.., 2Bg] row of h(t-) for position i of cat[src,dst], at element 2*offset + i */
  const int64_t* new_row;   /* [.., 2Bg] row of h(t'+) likewise */
  uint32_t* err;
  /* Partitioned state (owner != NULL): only nodes with owner[node] == my_rank are written (STEP 4-6 and the
   * event-before-memory check); the other positive nodes of the global batch belong to other ranks.  With
   * new_from_pending != 0 STEP 4 takes h(t'+) from tg_model.pending_vals (the owner's own table of precomputed
   * updater rows) instead of rows[new_row[..]], and new_row may be NULL. */
  const int32_t* owner;     /* [n_nodes] or NULL */
  int32_t my_rank;
  int32_t new_from_pending;
  /* Planned winners (upos != NULL): the latest-event-per-node dedup of the batch (select_latest_nids on float32 times,
   * tiger.py:232,419; memory.py:98) was made ahead - it depends on the batch only, not on node state - and only the
   * nodes this rank writes are listed: upos[p] the node, index[p] its winning position in cat[src, dst], *n_upos_dev the
   * count (device).  src / dst / eids are then the batch itself (offset_dev unused) and ts32 its float32 event times
   * tiled twice ([2 Bg]); the call is two launches (STEP 4+5 | STEP 6, or 4 | 5+6) with no dedup work. */
  const int64_t* upos;
  const int64_t* index;
  const int32_t* n_upos_dev;
  const float* ts32;
} tg_writeback_io;

/* What an owner serves and what a user adopts in the partitioned multi-GPU mode, one launch each.
 * tg_serve_rows: out[eff_pos[i], :] = effective right-memory row of eff_ids[i] and its time in column d
 * (tg_gather_eff_rows); out[msg_pos[j], :] = message-source memory row of msg_ids[j] (left memory for msg_src = left,
 * the effective right row otherwise) and its time.  out is [*, d + 1] float32.
 * tg_adopt_rows: the inverse on the receiving side - rows[eff_pos[i]] overwrites right_vals / right_ts of eff_ids[i],
 * rows[msg_pos[j]] the message-source memory row / time of msg_ids[j] (rows of nodes the rank does not own). */
int tg_serve_rows(const tg_model* m, int64_t n_eff, const int64_t* eff_ids, const int64_t* eff_pos, int64_t n_msg,
                  const int64_t* msg_ids, const int64_t* msg_pos, float* out, void* stream);
int tg_adopt_rows(const tg_model* m, int64_t n_eff, const int64_t* eff_ids, const int64_t* eff_pos, int64_t n_msg,
                  const int64_t* msg_ids, const int64_t* msg_pos, const float* rows, void* stream);

size_t tg_stream_writeback_workspace_bytes(const tg_model* m, int64_t Bg);
int tg_stream_writeback(const tg_model* m, const tg_writeback_io* io, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Multi-GPU, partitioned state: one global batch on one rank as ONE call     */
/* (www2023tiger_amd/dist.py: ResidentPartitionedStream; no reference          */
/* counterpart - per batch the owners' rows equal tiger.py:196-255 on the      */
/* global batch)                                                               */
/* ------------------------------------------------------------------------- */
/* The two exchanges of a step - PULL: owners -> users, the effective right-memory rows of remote involved nodes and the
 * message-source rows of remote "other" endpoints; PUSH: users -> owners, h(t-) of winning positions of nodes owned
 * elsewhere - are kernels that store straight into the PEER's window (memory every rank allocates with tg_xchg_alloc and
 * exports once with tg_ipc_export; peers map it with tg_ipc_import; ranks of one node: the stores travel over xGMI) and
 * raise an epoch flag there; the consuming kernels wait for the flags of all peers (bounded: TG_ERR_XCHG_TIMEOUT).  No
 * collective call, no host work inside a step: tg_part_step can be captured into a hipGraph, several steps per graph -
 * every per-step array below is a table over ALL steps of a resident stream, slice s = *step_dev, and the call ends by
 * advancing *step_dev.  Windows are double-buffered by step parity; every rank waits for every peer in every step, so a
 * rank is never more than one step ahead of a peer's reads.
 * Launches per step: begin (stage the plan slices, serve rows, signal; wait, arena mapping, adopt the pulled rows) | the
 * embedding step (tg_stream_step, embed_only + lean: sampler + centres, G, core, fc1, fc2; STEP 4 + 5 of the owner's own
 * winners ride on fc1's launch where it hosts riders) | push (rows, signal; else with STEP 4 + 5) | wait + STEP 6 (pushed rows
 * read in the window) | eager updater. */
#define TG_MAX_RANKS 16
typedef struct tg_part {
  int32_t world, rank;
  int64_t n_steps, Bg;      /* steps in the tables; events of a global batch */
  int64_t* step_dev;        /* device step counter (in / out) */
  int64_t* cur_step;        /* device scratch [1] */
  /* windows: pull_in[q] / push_in[q] / flags[q] = rank q's window as mapped HERE (q == rank: this rank's own).
   * pull inbox [2][world][pull_max] rows of d + 4 floats (row | time, 3 spare); push inbox [2][world][push_max] rows of d
   * floats; flags uint32 [4][TG_MAX_RANKS]: flags[kind][q] = last epoch (step + 1) rank q completed for `kind` (0 pull,
   * 1 push; 2, 3: tg_xchg_selftest) */
  float* pull_in[TG_MAX_RANKS];
  float* push_in[TG_MAX_RANKS];
  uint32_t* flags[TG_MAX_RANKS];
  int64_t pull_max, push_max;
  uint32_t* ticket;         /* device [4], zero: block counters of the two producing kernels, gate words of the two waits */
  uint32_t* err;            /* invariant word (TG_ERR_*) */
  /* plan tables (device; stride per step in brackets) */
  const int64_t *g_src, *g_dst, *g_eids;  /* the global batches [Bg] */
  const float* ts32;                      /* float32 event times tiled over cat[src, dst] [2 Bg] */
  const int64_t* left_row;                /* row of h(t-) per position of cat[src, dst] [2 Bg]: own rows of the rank's
                                           * [3B | world * push_max] output buffer, received rows behind them */
  int64_t serve_cap;                      /* PULL, owner side [serve_cap]: n_serve[s] live entries */
  const int32_t *n_serve, *serve_row, *serve_kind, *serve_peer, *serve_slot;
  const int32_t *adopt_row, *adopt_kind;  /* PULL, user side [world * pull_max]: inbox slot -> state row (-1: unused), kind */
  int64_t req_cap;                        /* arena mapping [req_cap]: row_of[req_node] = req_row for this step; */
  const int32_t* n_req;                   /* row_of[unmap_node] = -1 first: the previous step's pulled nodes that this */
  const int64_t* req_node;                /* step does not pull (their arena rows hold other nodes from now on)        */
  const int32_t* req_row;
  const int32_t* n_unmap;
  const int64_t* unmap_node;
  int64_t push_cap;                       /* PUSH, user side [push_cap] */
  const int32_t *n_push, *push_src, *push_peer, *push_slot;
  int64_t mine_cap;                       /* the winners this rank writes [mine_cap]: node, position, state row */
  const int32_t* n_mine;
  const int64_t *mine_node, *mine_index, *mine_row;
  /* staging (device, fixed addresses): this step's slices as the write-back / updater launches read them */
  int64_t *st_src, *st_dst, *st_eids, *st_left_row, *st_mine_node, *st_mine_index, *st_mine_row;
  float* st_ts32;
  int32_t *st_mine32, *st_n_mine;
  const int32_t* owner;     /* [n_nodes] */
  int32_t* row_of;          /* the model's row_of, writable (NULL: full-height tables) */
} tg_part;
/* io: an embed_only + lean step over the rank's resident events (h with room for 3 B + world * push_max rows);
 * ws: its workspace; aws: tg_apply_messages_workspace_bytes(m, mine_cap). */
int tg_part_step(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, const tg_part* p, void* ws, size_t ws_bytes,
                 void* aws, size_t aws_bytes, void* stream);
/* `rounds` ping rounds through the mapped windows with the step's own store / load / flag forms (collective: every rank
 * calls it; flags of kind 2 / 3, the push inboxes' first slots): *result_dev (device int32, zeroed by the caller) stays 0
 * when every word every peer stored here arrived - bit 0: a flag timed out, bit 1: a wrong word, bit 2: the second
 * hand-shake timed out.  The caller zeroes the window afterwards. */
int tg_xchg_selftest(const tg_part* p, int32_t d, int32_t rounds, int32_t* result_dev, void* stream);
int tg_xchg_alloc(size_t bytes, void** out);  /* zero-filled device memory peers may store into */
int tg_xchg_clear(void* p, size_t bytes);      /* zero it again (synchronous) */
int tg_xchg_free(void* p);
int tg_ipc_export(void* p, uint8_t* handle64);            /* hipIpcGetMemHandle */
int tg_ipc_import(const uint8_t* handle64, void** out);   /* hipIpcOpenMemHandle (another process's window) */
int tg_ipc_close(void* p);

#ifdef __cplusplus
}
#endif
#endif /* TIGER_HIP_H */
