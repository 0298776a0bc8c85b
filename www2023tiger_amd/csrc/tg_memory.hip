// Node-memory / mailbox gather-scatter kernels (SURVEY.md K5, K7, K9, K10, K11-static, K12;
// a11-a13, a16, a17, a19).  Reference: tiger/model/memory.py:12-138,
// tiger/model/tiger.py:229-255,396-442,594-609, tiger/model/time_encoding.py:24-26.
// All HBM-bound: rows are moved as float4 (16 B per lane), one wavefront per row where a
// row also carries scalar state (timestamp, has-message bit), flat otherwise.
#include <algorithm>

#include "tg_common.h"
#include "tg_part.h"

namespace tg {

__global__ void k_time_encode(int64_t n, const float* __restrict__ ts, int d, const float* __restrict__ freq,
                              const float* __restrict__ phase, float* __restrict__ out) {
  const int64_t total = n * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / d;
    const int c = (int)(t - i * d);
    out[t] = time_enc(ts[i], freq[c], phase[c]);
  }
}

__global__ void k_gather_rows(int64_t n, const int32_t* __restrict__ n_dev, const int64_t* __restrict__ ids, int w4,
                              const float4* __restrict__ table, float4* __restrict__ out,
                              const float* __restrict__ ts_table, float* __restrict__ ts_out) {
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int64_t total = n * w4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / w4;
    const int c = (int)(t - i * w4);
    const int64_t id = ids[i];
    out[t] = table[id * w4 + c];
    if (c == 0 && ts_out) ts_out[i] = ts_table[id];
  }
}

__global__ void k_memory_scatter(int64_t n, const int32_t* __restrict__ n_dev, const int64_t* __restrict__ ids,
                                 const int64_t* __restrict__ src_index, const int64_t* __restrict__ ts_index, int w4,
                                 const float4* __restrict__ vals,
                                 const float* __restrict__ ts, float4* __restrict__ table, float* __restrict__ ts_table,
                                 uint8_t* __restrict__ active, int check, uint32_t* __restrict__ err) {
  if (n_dev) n = min(n, (int64_t)*n_dev);
  const int64_t total = n * w4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / w4;
    const int c = (int)(t - i * w4);
    const int64_t id = ids[i];
    const int64_t s = src_index ? src_index[i] : i;
    table[id * w4 + c] = vals[s * w4 + c];
    if (c == 0) {
      const float nt = ts[ts_index ? ts_index[i] : s];
      if (check && ts_table[id] > nt) atomicOr(err, TG_ERR_PAST_MEMORY);
      ts_table[id] = nt;
      if (active) active[id] = 1;
    }
  }
}

// STEP 4 (tiger.py:230-241): one wavefront per unique positive node.
__global__ void __launch_bounds__(256) k_consume_update_right(tg_model m, const int64_t* __restrict__ upos,
                                                              const int32_t* __restrict__ n_upos, int64_t cap,
                                                              const float4* __restrict__ reprs,
                                                              const uint64_t* __restrict__ bm,
                                                              const uint32_t* __restrict__ rank,
                                                              const int64_t* __restrict__ row_index,
                                                              uint32_t* __restrict__ err) {
  const int64_t n = min((int64_t)*n_upos, cap);
  const int lane = lane_id();
  const int w4 = m.d / 4;
  float4* right = reinterpret_cast<float4*>(m.right_vals);
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += (int64_t)gridDim.x * 4) {
    const int64_t id = upos[p];
    if (!bm_test(m.has_msg, id)) continue;  // not outdated: nothing to consume (wave-uniform)
    const int64_t u = row_index ? row_index[p] : (int64_t)bm_rank(bm, rank, id);
    for (int c = lane; c < w4; c += TG_WAVE) right[id * w4 + c] = reprs[u * w4 + c];
    if (lane == 0) {
      const float mts = m.msg_ts[id];
      if (m.right_ts[id] > mts) atomicOr(err, TG_ERR_PAST_MEMORY);
      m.right_ts[id] = mts;
      if (m.right_active) m.right_active[id] = 1;
      atomicAnd((unsigned long long*)(m.has_msg + (id >> 6)), ~(1ull << (id & 63)));
    }
  }
}

// STEP 5 (tiger.py:422-442, memory.py:77-106): one wavefront per unique positive node
// builds [own | other | edge | time] from the message memory and writes the mailbox row.
// Waves additionally check the "event precedes memory" invariant over all 2B positions.
__global__ void __launch_bounds__(256) k_store_events(tg_model m, int64_t B, const int64_t* __restrict__ src,
                                                      const int64_t* __restrict__ dst, const float* __restrict__ ts,
                                                      const int64_t* __restrict__ eids, const int64_t* __restrict__ upos,
                                                      const int64_t* __restrict__ index,
                                                      const int32_t* __restrict__ n_upos, uint32_t* __restrict__ err) {
  const int64_t n = min((int64_t)*n_upos, 2 * B);
  const int lane = lane_id();
  const int d4 = m.d / 4, e4 = m.d_e / 4;
  const int row4 = 3 * d4 + e4;
  const float* mem_ts = (m.msg_src == TG_SRC_LEFT) ? m.left_ts : m.right_ts;
  const float4* mem = reinterpret_cast<const float4*>((m.msg_src == TG_SRC_LEFT) ? m.left_vals : m.right_vals);
  const float4* nf = reinterpret_cast<const float4*>(m.nfeats);
  const float4* ef = reinterpret_cast<const float4*>(m.efeats);
  const float4* fq = reinterpret_cast<const float4*>(m.te_freq);
  const float4* ph = reinterpret_cast<const float4*>(m.te_phase);
  float4* box = reinterpret_cast<float4*>(m.msg_vals);
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwave = (int64_t)gridDim.x * 4;
  for (int64_t i = wave0 * TG_WAVE + lane; i < 2 * B; i += nwave * TG_WAVE) {
    const int64_t e = i < B ? i : i - B;
    const int64_t node = i < B ? src[e] : dst[e];
    if (mem_ts[node] > ts[e]) atomicOr(err, TG_ERR_EVENT_BEFORE_MEM);
  }
  for (int64_t p = wave0; p < n; p += nwave) {
    const int64_t own = upos[p];
    const int64_t idx = index[p];
    const int64_t e = idx < B ? idx : idx - B;
    const int64_t other = idx < B ? dst[e] : src[e];
    const float t = ts[e];
    const float dt = t - mem_ts[own];
    const int64_t eid = eids[e];
    for (int c = lane; c < row4; c += TG_WAVE) {
      float4 v;
      if (c < 2 * d4) {
        const int64_t node = c < d4 ? own : other;
        const int cc = c < d4 ? c : c - d4;
        v = mem[node * d4 + cc];
        if (nf) {
          const float4 f = nf[node * d4 + cc];
          v.x += f.x; v.y += f.y; v.z += f.z; v.w += f.w;
        }
      } else if (c < 2 * d4 + e4) {
        v = ef ? ef[eid * e4 + (c - 2 * d4)] : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int cc = c - 2 * d4 - e4;
        const float4 w = fq[cc], q = ph[cc];
        v = make_float4(time_enc(dt, w.x, q.x), time_enc(dt, w.y, q.y), time_enc(dt, w.z, q.z),
                        time_enc(dt, w.w, q.w));
      }
      box[own * row4 + c] = v;
    }
    if (lane == 0) {
      const uint64_t bit = 1ull << (own & 63);
      const unsigned long long old = atomicOr((unsigned long long*)(m.has_msg + (own >> 6)), bit);
      if (old & bit) atomicOr(err, TG_ERR_UNUSED_MESSAGE);
      m.msg_ts[own] = t;
    }
  }
}

// reprs[u] = right_memory[involved[u]] (tiger.py:214) fused with the invariants of compute_messages
// (message_modules.py:158-159, tiger.py:325-327) over the outdated list.  Rows of nodes with a pending
// message: lazy form (EAGER = false) - NOT copied, the updater launch that follows writes h(t'+) to exactly
// those rows of reprs (tiger.py:219-221), copying them would be dead stores (in steady state most of the
// involved set); eager form - gathered from the table of precomputed updater rows (tg_model.pending_vals), which
// makes STEP 1-2 this one gather.
template <bool EAGER>
__global__ void k_consume_gather_check(tg_model m, const int64_t* __restrict__ involved,
                                       const int32_t* __restrict__ n_involved, int64_t cap, float4* __restrict__ reprs,
                                       const int64_t* __restrict__ outdated, const int32_t* __restrict__ n_outdated,
                                       uint32_t* __restrict__ err, PosArgs pos) {
  const int w4 = m.d / 4;
  const int64_t n = min((int64_t)*n_involved, cap);
  const int64_t total = n * w4;
  const float4* right = reinterpret_cast<const float4*>(m.right_vals);
  const float4* pend = reinterpret_cast<const float4*>(m.pending_vals);
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = tid; t < total; t += nth) {
    const int64_t i = t / w4;
    const int64_t id = involved[i];
    const bool pending = bm_test(m.has_msg, id);
    if (EAGER) {
      reprs[t] = (pending ? pend : right)[id * w4 + (t - i * w4)];
    } else {
      if (pending) continue;
      reprs[t] = right[id * w4 + (t - i * w4)];
    }
  }
  const int64_t no = min((int64_t)*n_outdated, cap);
  const float* mem_ts = (m.msg_src == TG_SRC_LEFT) ? m.left_ts : m.right_ts;
  for (int64_t i = tid; i < no; i += nth) {
    const int64_t id = outdated[i];
    const float mts = m.msg_ts[id], last = mem_ts[id];
    if (last > mts) atomicOr(err, TG_ERR_MSG_BEFORE_MEM);
    if (m.msg_src == TG_SRC_LEFT && !(mts == last)) atomicOr(err, TG_ERR_MSG_TS_MISMATCH);
  }
  if (pos.best) pos_max_pass(pos, tid, nth);
}

template <int PHASE>
__global__ void __launch_bounds__(256) k_writeback(tg_model m, WritebackArgs a) {
  writeback_body<PHASE>(m, a, blockIdx.x, gridDim.x);
}

// ---- tg_part_step (tg_part.hip): the two write-back launches of a rank's own winners with the PUSH exchange folded in
__global__ void __launch_bounds__(256) k_part_push_wb0(tg_model m, WritebackArgs a, tg_part p, const float* __restrict__ h,
                                                       unsigned push_blocks) {
  if (blockIdx.x < push_blocks) {  // h(t-) of the winning positions of nodes owned elsewhere -> their owners' windows
    const int64_t s = p.cur_step[0];
    if (s < p.n_steps) {
      const int w4 = m.d / 4;
      const int64_t n = p.n_push[s], par = s & 1;
      const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)push_blocks * blockDim.x;
      for (int64_t t = tid; t < n * w4; t += nth) {
        const int64_t i = t / w4;
        const int c = (int)(t - i * w4);
        const int64_t e = s * p.push_cap + i;
        float4* inbox = reinterpret_cast<float4*>(p.push_in[p.push_peer[e]]) +
                        ((par * p.world + p.rank) * p.push_max + p.push_slot[e]) * w4;
        st_sys(inbox + c, reinterpret_cast<const float4*>(h)[(int64_t)p.push_src[e] * w4 + c]);
      }
    }
    signal_peers(p, 1, (uint32_t)(s + 1), push_blocks);
    return;
  }
  if (p.cur_step[0] < p.n_steps) writeback_body<0>(m, a, blockIdx.x - push_blocks, gridDim.x - push_blocks);
}
__global__ void __launch_bounds__(256) k_part_wb1(tg_model m, WritebackArgs a, tg_part p, int64_t hi_from) {
  const int64_t s = p.cur_step[0];
  if (s >= p.n_steps) return;
  wait_peers(p, 1, (uint32_t)(s + 1), blockIdx.x);
  const int64_t slots = (int64_t)p.world * p.push_max;
  const float4* inbox = reinterpret_cast<const float4*>(p.push_in[p.rank]) + (s & 1) * slots * (m.d / 4);
  writeback_body<1>(m, a, blockIdx.x, gridDim.x, inbox, hi_from);
  if (blockIdx.x == 0 && threadIdx.x == 0) *p.step_dev = s + 1;  // (every launch of this step reads cur_step)
}
int part_push_wb0_launch(const tg_model* m, const WritebackArgs& a, const tg_part* p, const float* h, bool with_wb0,
                         hipStream_t st) {
  const unsigned pb = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, cdiv(p->push_cap * (m->d / 4), 256)));
  hipLaunchKernelGGL(k_part_push_wb0, dim3(pb + (with_wb0 ? flat_grid(2 * a.B, 4) : 0u)), dim3(256), 0, st, *m, a, *p, h, pb);
  return check_launch("tg_part_step(push + write-back 0)");
}
int part_wb1_launch(const tg_model* m, const WritebackArgs& a, const tg_part* p, int64_t hi_from, hipStream_t st) {
  hipLaunchKernelGGL(k_part_wb1, dim3(flat_grid(2 * a.B, 4)), dim3(256), 0, st, *m, a, *p, hi_from);
  return check_launch("tg_part_step(write-back 1)");
}

// ---- STEP 4-6 in ONE launch (eager updates, direct form): writeback_fused_body (tg_common.h), which can also ride on
// the launch of the attention block's last product (WbRider)
__global__ void __launch_bounds__(256) k_writeback_fused(tg_model m, WritebackArgs a) {
  writeback_fused_body<true>(m, a, blockIdx.x, gridDim.x);
}

int consume_gather_check_launch(const tg_model* m, const int64_t* involved, const int32_t* n_involved, int64_t cap,
                                float* reprs, const int64_t* outdated, const int32_t* n_outdated, uint32_t* err,
                                hipStream_t st, const PosArgs* pos, bool eager) {
  if (eager && !m->pending_vals) return TG_EINVAL;
  if (eager)
    TG_KLAUNCH(k_consume_gather_check<true>, dim3(flat_grid(cap * (m->d / 4), 256)), dim3(256), 0, st, *m,
                       involved, n_involved, cap, (float4*)reprs, outdated, n_outdated, err, pos ? *pos : PosArgs{});
  else
    TG_KLAUNCH(k_consume_gather_check<false>, dim3(flat_grid(cap * (m->d / 4), 256)), dim3(256), 0, st, *m,
                       involved, n_involved, cap, (float4*)reprs, outdated, n_outdated, err, pos ? *pos : PosArgs{});
  return check_launch("consume_gather_check");
}

int writeback_launch(const tg_model* m, const WritebackArgs& a, int phase, hipStream_t st) {
  const unsigned grid = flat_grid(2 * a.B, 4);
  if (phase == 2 && (!a.snap || !a.snap_ts || !m->pending_vals || a.rows || a.owner)) return TG_EINVAL;
  if (phase == 0)
    TG_KLAUNCH(k_writeback<0>, dim3(grid), dim3(256), 0, st, *m, a);
  else if (phase == 1)
    TG_KLAUNCH(k_writeback<1>, dim3(grid), dim3(256), 0, st, *m, a);
  else
    TG_KLAUNCH(k_writeback_fused, dim3(grid), dim3(256), 0, st, *m, a);
  return check_launch("writeback");
}

// TIGER.restart state update (tiger.py:603,608-609)
__global__ void k_restart_apply(tg_model m, int64_t n, const int64_t* __restrict__ nids,
                                const float4* __restrict__ hl, const float4* __restrict__ hr,
                                const float* __restrict__ pt, const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min(n, (int64_t)*n_dev);  // (a capacity-sized launch inside a captured graph: the live count is on the device)
  const int w4 = m.d / 4;
  const int64_t total = n * w4;
  float4* left = reinterpret_cast<float4*>(m.left_vals);
  float4* right = reinterpret_cast<float4*>(m.right_vals);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / w4;
    const int c = (int)(t - i * w4);
    const int64_t id = nids[i];
    left[id * w4 + c] = hl[t];
    right[id * w4 + c] = hr[t];
    if (c == 0) {
      m.left_ts[id] = pt[i];
      m.right_ts[id] = pt[i];
      if (m.left_active) m.left_active[id] = 1;
      if (m.right_active) m.right_active[id] = 1;
      atomicAnd((unsigned long long*)(m.has_msg + (id >> 6)), ~(1ull << (id & 63)));
    }
  }
}

}  // namespace tg

using namespace tg;

static int model_ok(const tg_model* m) {
  return m && m->n_nodes > 0 && m->d > 0 && (m->d % 4) == 0 && m->d_e > 0 && (m->d_e % 4) == 0;
}

extern "C" int tg_time_encode(int64_t n, const float* ts, int32_t d, const float* freq, const float* phase, float* out,
                              void* stream) {
  if (n < 0 || d <= 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!ts || !freq || !phase || !out) return TG_EINVAL;
  TG_KLAUNCH(k_time_encode, dim3(flat_grid(n * d, 256)), dim3(256), 0, as_stream(stream), n, ts, d, freq,
                     phase, out);
  return check_launch("tg_time_encode");
}

extern "C" int tg_gather_rows(int64_t n, const int64_t* ids, int32_t width, const float* table, float* out,
                              const float* ts_table, float* ts_out, void* stream) {
  if (n < 0 || width <= 0 || (width % 4) != 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!ids || !table || !out || (ts_out && !ts_table)) return TG_EINVAL;
  TG_KLAUNCH(k_gather_rows, dim3(flat_grid(n * (width / 4), 256)), dim3(256), 0, as_stream(stream), n,
                     (const int32_t*)nullptr, ids, width / 4, (const float4*)table, (float4*)out, ts_table, ts_out);
  return check_launch("tg_gather_rows");
}

extern "C" int tg_memory_scatter(int64_t n, const int32_t* n_dev, const int64_t* ids, const int64_t* src_index,
                                 int32_t width, const float* vals, const float* ts, float* table, float* ts_table,
                                 uint8_t* active, int32_t check, uint32_t* err, void* stream) {
  if (n < 0 || width <= 0 || (width % 4) != 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!ids || !vals || !ts || !table || !ts_table || (check && !err)) return TG_EINVAL;
  TG_KLAUNCH(k_memory_scatter, dim3(flat_grid(n * (width / 4), 256)), dim3(256), 0, as_stream(stream), n, n_dev,
                     ids, src_index, (const int64_t*)nullptr, width / 4, (const float4*)vals, ts, (float4*)table, ts_table,
                     active, check, err);
  return check_launch("tg_memory_scatter");
}

extern "C" int tg_memory_scatter2(int64_t n, const int32_t* n_dev, const int64_t* ids, const int64_t* val_index,
                                  const int64_t* ts_index, int32_t width, const float* vals, const float* ts,
                                  float* table, float* ts_table, uint8_t* active, int32_t check, uint32_t* err,
                                  void* stream) {
  if (n < 0 || width <= 0 || (width % 4) != 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!ids || !vals || !ts || !table || !ts_table || (check && !err)) return TG_EINVAL;
  TG_KLAUNCH(k_memory_scatter, dim3(flat_grid(n * (width / 4), 256)), dim3(256), 0, as_stream(stream), n, n_dev,
                     ids, val_index, ts_index, width / 4, (const float4*)vals, ts, (float4*)table, ts_table, active, check,
                     err);
  return check_launch("tg_memory_scatter2");
}

extern "C" int tg_mailbox_consume_gather(const tg_model* m, const int64_t* involved, const int32_t* n_involved,
                                         int64_t cap, float* reprs, void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!model_ok(m) || cap < 0) return TG_EINVAL;
  if (cap == 0) return TG_OK;
  if (!involved || !n_involved || !reprs) return TG_EINVAL;
  TG_KLAUNCH(k_gather_rows, dim3(flat_grid(cap * (m->d / 4), 256)), dim3(256), 0, as_stream(stream), cap,
                     n_involved, involved, m->d / 4, (const float4*)m->right_vals, (float4*)reprs,
                     (const float*)nullptr, (float*)nullptr);
  return check_launch("tg_mailbox_consume_gather");
}

extern "C" int tg_consume_update_right(const tg_model* m, const int64_t* upos, const int32_t* n_upos, int64_t cap,
                                       const float* reprs, const uint64_t* bitmap, const uint32_t* rank, uint32_t* err,
                                       void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!model_ok(m) || cap < 0) return TG_EINVAL;
  if (cap == 0) return TG_OK;
  if (!upos || !n_upos || !reprs || !bitmap || !rank || !err) return TG_EINVAL;
  TG_KLAUNCH(k_consume_update_right, dim3(flat_grid(cap, 4)), dim3(256), 0, as_stream(stream), *m, upos, n_upos,
                     cap, (const float4*)reprs, bitmap, rank, (const int64_t*)nullptr, err);
  return check_launch("tg_consume_update_right");
}

extern "C" int tg_consume_update_right_rows(const tg_model* m, const int64_t* upos, const int32_t* n_upos, int64_t cap,
                                            const float* rows, const int64_t* row_index, uint32_t* err, void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!model_ok(m) || cap < 0) return TG_EINVAL;
  if (cap == 0) return TG_OK;
  if (!upos || !n_upos || !rows || !row_index || !err) return TG_EINVAL;
  TG_KLAUNCH(k_consume_update_right, dim3(flat_grid(cap, 4)), dim3(256), 0, as_stream(stream), *m, upos, n_upos,
                     cap, (const float4*)rows, (const uint64_t*)nullptr, (const uint32_t*)nullptr, row_index, err);
  return check_launch("tg_consume_update_right_rows");
}

namespace tg {
__global__ void k_gather_eff_rows(tg_model m, int64_t n, const int64_t* __restrict__ ids, float4* __restrict__ out,
                                  float* __restrict__ ts_out) {
  const int w4 = m.d / 4;
  const float4* right = reinterpret_cast<const float4*>(m.right_vals);
  const float4* pend = reinterpret_cast<const float4*>(m.pending_vals);
  const int64_t total = n * w4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / w4;
    const int c = (int)(t - i * w4);
    const int64_t id = ids[i];
    const bool pending = bm_test(m.has_msg, id);
    out[t] = (pending ? pend : right)[id * w4 + c];
    if (c == 0 && ts_out) ts_out[i] = pending ? m.msg_ts[id] : m.right_ts[id];
  }
}
}  // namespace tg

namespace tg {
// out[pos[i], 0..d) = row, out[pos[i], d] = time; kind 0: effective right memory, kind 1: left memory
__global__ void k_serve_rows(tg_model m, int64_t n_eff, const int64_t* __restrict__ eff_ids,
                             const int64_t* __restrict__ eff_pos, int64_t n_msg, const int64_t* __restrict__ msg_ids,
                             const int64_t* __restrict__ msg_pos, float* __restrict__ out) {
  const int d = m.d, w4 = d / 4, ld = d + 1;
  const float4* right = reinterpret_cast<const float4*>(m.right_vals);
  const float4* pend = reinterpret_cast<const float4*>(m.pending_vals);
  const float4* left = reinterpret_cast<const float4*>(m.left_vals);
  const bool msg_left = m.msg_src == TG_SRC_LEFT;
  const int64_t total = (n_eff + n_msg) * w4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / w4;
    const int c = (int)(t - i * w4);
    const bool is_msg = i >= n_eff;
    const int64_t id = is_msg ? msg_ids[i - n_eff] : eff_ids[i];
    const int64_t row = is_msg ? msg_pos[i - n_eff] : eff_pos[i];
    float4 v;
    float ts;
    if (is_msg && msg_left) {
      v = left[id * w4 + c];
      ts = m.left_ts[id];
    } else {
      const bool pending = bm_test(m.has_msg, id);
      v = (pending ? pend : right)[id * w4 + c];
      ts = pending ? m.msg_ts[id] : m.right_ts[id];
    }
    float* o = out + row * ld + 4 * c;  // rows of d + 1 floats are not 16-byte aligned
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    if (c == 0) out[row * ld + d] = ts;
  }
}
__global__ void k_adopt_rows(tg_model m, int64_t n_eff, const int64_t* __restrict__ eff_ids,
                             const int64_t* __restrict__ eff_pos, int64_t n_msg, const int64_t* __restrict__ msg_ids,
                             const int64_t* __restrict__ msg_pos, const float* __restrict__ rows) {
  const int d = m.d, w4 = d / 4, ld = d + 1;
  const bool msg_left = m.msg_src == TG_SRC_LEFT;
  const int64_t total = (n_eff + n_msg) * w4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / w4;
    const int c = (int)(t - i * w4);
    const bool is_msg = i >= n_eff;
    const int64_t id = is_msg ? msg_ids[i - n_eff] : eff_ids[i];
    const int64_t row = is_msg ? msg_pos[i - n_eff] : eff_pos[i];
    const float* r = rows + row * ld + 4 * c;
    const float4 v = make_float4(r[0], r[1], r[2], r[3]);
    const bool to_left = is_msg && msg_left;
    reinterpret_cast<float4*>(to_left ? m.left_vals : m.right_vals)[id * w4 + c] = v;
    if (c == 0) (to_left ? m.left_ts : m.right_ts)[id] = rows[row * ld + d];
  }
}
}  // namespace tg

extern "C" int tg_serve_rows(const tg_model* m, int64_t n_eff, const int64_t* eff_ids, const int64_t* eff_pos,
                             int64_t n_msg, const int64_t* msg_ids, const int64_t* msg_pos, float* out, void* stream) {
  if (!model_ok(m) || n_eff < 0 || n_msg < 0 || !m->pending_vals) return TG_EINVAL;
  if (n_eff + n_msg == 0) return TG_OK;
  if (!out || (n_eff && (!eff_ids || !eff_pos)) || (n_msg && (!msg_ids || !msg_pos))) return TG_EINVAL;
  TG_KLAUNCH(k_serve_rows, dim3(flat_grid((n_eff + n_msg) * (m->d / 4), 256)), dim3(256), 0, as_stream(stream), *m,
                     n_eff, eff_ids, eff_pos, n_msg, msg_ids, msg_pos, out);
  return check_launch("tg_serve_rows");
}

extern "C" int tg_adopt_rows(const tg_model* m, int64_t n_eff, const int64_t* eff_ids, const int64_t* eff_pos,
                             int64_t n_msg, const int64_t* msg_ids, const int64_t* msg_pos, const float* rows,
                             void* stream) {
  if (!model_ok(m) || n_eff < 0 || n_msg < 0) return TG_EINVAL;
  if (n_eff + n_msg == 0) return TG_OK;
  if (!rows || (n_eff && (!eff_ids || !eff_pos)) || (n_msg && (!msg_ids || !msg_pos))) return TG_EINVAL;
  TG_KLAUNCH(k_adopt_rows, dim3(flat_grid((n_eff + n_msg) * (m->d / 4), 256)), dim3(256), 0, as_stream(stream), *m,
                     n_eff, eff_ids, eff_pos, n_msg, msg_ids, msg_pos, rows);
  return check_launch("tg_adopt_rows");
}

extern "C" int tg_gather_eff_rows(const tg_model* m, int64_t n, const int64_t* ids, float* out, float* ts_out,
                                  void* stream) {
  if (!model_ok(m) || n < 0 || !m->pending_vals) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!ids || !out) return TG_EINVAL;
  TG_KLAUNCH(k_gather_eff_rows, dim3(flat_grid(n * (m->d / 4), 256)), dim3(256), 0, as_stream(stream), *m, n, ids,
                     (float4*)out, ts_out);
  return check_launch("tg_gather_eff_rows");
}

extern "C" int tg_store_events(const tg_model* m, int64_t B, const int64_t* src, const int64_t* dst, const float* ts,
                               const int64_t* eids, const int64_t* upos, const int64_t* index, const int32_t* n_upos,
                               uint32_t* err, void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!model_ok(m) || B < 0) return TG_EINVAL;
  if (B == 0) return TG_OK;
  if (!src || !dst || !ts || !eids || !upos || !index || !n_upos || !err) return TG_EINVAL;
  TG_KLAUNCH(k_store_events, dim3(flat_grid(2 * B, 4)), dim3(256), 0, as_stream(stream), *m, B, src, dst, ts,
                     eids, upos, index, n_upos, err);
  return check_launch("tg_store_events");
}

extern "C" int tg_restart_apply(const tg_model* m, int64_t n, const int64_t* nids, const float* h_left,
                                const float* h_right, const float* prev_ts, void* stream) {
  if (m && m->row_of) return TG_EUNSUPPORTED;  // state addressed by node id: not on physically partitioned tables (tg_model.row_of)
  if (!model_ok(m) || n < 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  if (!nids || !h_left || !h_right || !prev_ts) return TG_EINVAL;
  return tg::restart_apply_dev(m, n, nids, h_left, h_right, prev_ts, nullptr, as_stream(stream));
}
int tg::restart_apply_dev(const tg_model* m, int64_t n, const int64_t* nids, const float* h_left, const float* h_right,
                          const float* prev_ts, const int32_t* n_dev, hipStream_t st) {
  TG_KLAUNCH(k_restart_apply, dim3(flat_grid(n * (m->d / 4), 256)), dim3(256), 0, st, *m, n, nids, (const float4*)h_left,
             (const float4*)h_right, prev_ts, n_dev);
  return check_launch("tg_restart_apply");
}
