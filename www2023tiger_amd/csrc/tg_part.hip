// Partitioned multi-GPU step (include/tiger_hip.h: tg_part): one global batch on one rank as ONE call that a hipGraph can
// hold - the two exchanges of rows between ranks are kernels of this library that store straight into the peer's window
// (memory every rank exports once, hipIpcGetMemHandle; on one node the stores travel over xGMI) and signal with epoch
// flags; there is no collective call, hence no host work, inside a step.  The plan of every step - which rows travel
// where; a function of the graph and the stream only (www2023tiger_amd/dist.py) - sits in device tables indexed by a
// device-side step counter, so a captured graph of g steps replays g consecutive global batches.
//
// Reference: there is none for this layout (its multi-GPU mode is time-chunk DDP, train_self_supervised_ddp.py:145-211);
// per batch the owners' rows must equal the single-GPU engine's on the global batch (tiger.py:196-255), which
// tests/test_dist.py checks.
#include <string.h>

#include "tg_part.h"
#include "tg_step.h"

namespace tg {

// ---- launch 1: stage this step's plan slices at fixed addresses (the write-back / updater launches of the single-GPU
// engine read them there), forget the previous step's arena mappings, serve the rows peers pull, signal; then wait for the
// rows this rank pulls and adopt them.  The grid is at most 256 workgroups: all resident at once.
__global__ void __launch_bounds__(256) k_part_begin(tg_model m, tg_part p) {
  const int64_t s = *p.step_dev;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  const int64_t Bg = p.Bg;
  if (s < p.n_steps) {
    for (int64_t i = tid; i < Bg; i += nth) {
      p.st_src[i] = p.g_src[s * Bg + i];
      p.st_dst[i] = p.g_dst[s * Bg + i];
      p.st_eids[i] = p.g_eids[s * Bg + i];
    }
    for (int64_t i = tid; i < 2 * Bg; i += nth) {
      p.st_ts32[i] = p.ts32[s * 2 * Bg + i];
      p.st_left_row[i] = p.left_row[s * 2 * Bg + i];
    }
    const int nm = p.n_mine[s];
    for (int64_t i = tid; i < nm; i += nth) {
      p.st_mine_node[i] = p.mine_node[s * p.mine_cap + i];
      p.st_mine_index[i] = p.mine_index[s * p.mine_cap + i];
      const int64_t r = p.mine_row[s * p.mine_cap + i];
      p.st_mine_row[i] = r;
      p.st_mine32[i] = (int32_t)r;
    }
    if (tid == 0) *p.st_n_mine = nm;
    if (p.row_of) {  // the previous batch's pulled nodes that this batch does not pull: their arena rows hold other nodes now
      const int np = p.n_unmap[s];
      for (int64_t i = tid; i < np; i += nth) p.row_of[p.unmap_node[s * p.req_cap + i]] = -1;
    }
    // PULL, owner side: row -> the requester's inbox (this rank's block, the slot the plan agreed on)
    const int d = m.d, w4 = d / 4, ld4 = w4 + 1;  // inbox rows: d floats + [time, 3 spare] = d / 4 + 1 float4
    const int64_t n = p.n_serve[s];
    const float4* right = reinterpret_cast<const float4*>(m.right_vals);
    const float4* pend = reinterpret_cast<const float4*>(m.pending_vals);
    const float4* left = reinterpret_cast<const float4*>(m.left_vals);
    const bool msg_left = m.msg_src == TG_SRC_LEFT;
    const int64_t par = s & 1;
    for (int64_t t = tid; t < n * ld4; t += nth) {
      const int64_t i = t / ld4;
      const int c = (int)(t - i * ld4);
      const int64_t e = s * p.serve_cap + i;
      const int64_t row = p.serve_row[e];
      const bool from_left = p.serve_kind[e] != 0 && msg_left;
      const bool pending = !from_left && bm_test(m.has_msg, row);
      float4 v;
      if (c < w4) v = (from_left ? left : (pending ? pend : right))[row * w4 + c];
      else v = make_float4(from_left ? m.left_ts[row] : (pending ? m.msg_ts[row] : m.right_ts[row]), 0.f, 0.f, 0.f);
      float4* inbox = reinterpret_cast<float4*>(p.pull_in[p.serve_peer[e]]) +
                      ((par * p.world + p.rank) * p.pull_max + p.serve_slot[e]) * ld4;
      st_sys(inbox + c, v);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) p.cur_step[0] = s;  // what the later launches of this step read
  signal_peers(p, 0, (uint32_t)(s + 1), gridDim.x);
  if (s >= p.n_steps) return;
  // ---- the same launch, user side: wait for the pulled rows (every rank serves before it waits, and the whole grid is
  // resident: no rank's serving workgroups wait for anything), point row_of at this batch's arena rows, adopt the rows
  wait_peers(p, 0, (uint32_t)(s + 1), blockIdx.x);
  if (p.row_of) {
    const int nr = p.n_req[s];
    for (int64_t i = tid; i < nr; i += nth) p.row_of[p.req_node[s * p.req_cap + i]] = p.req_row[s * p.req_cap + i];
  }
  const int d = m.d, w4 = d / 4, ld4 = w4 + 1;
  const bool msg_left = m.msg_src == TG_SRC_LEFT;
  const int64_t slots = (int64_t)p.world * p.pull_max, par = s & 1;
  const float4* inbox = reinterpret_cast<const float4*>(p.pull_in[p.rank]) + par * slots * ld4;
  for (int64_t t = tid; t < slots * w4; t += nth) {
    const int64_t i = t / w4;
    const int c = (int)(t - i * w4);
    const int64_t row = p.adopt_row[s * slots + i];
    if (row < 0) continue;
    const bool to_left = p.adopt_kind[s * slots + i] != 0 && msg_left;
    reinterpret_cast<float4*>(to_left ? m.left_vals : m.right_vals)[row * w4 + c] = ld_sys(inbox + i * ld4 + c);
    if (c == 0) (to_left ? m.left_ts : m.right_ts)[row] = ld_sys(inbox + i * ld4 + w4).x;
  }
}

// ---- self-test of the windows (before a stream is driven through them): `rounds` ping rounds, one workgroup per rank.
// Round r: lane q stores (rank << 16 | r) into peer q's push inbox (this rank's block, slot 0, parity r & 1) with the
// step's own store form, the rank's flag of kind 2 goes up; after the wait for every peer's flag lane q checks what peer q
// stored here; a second hand-shake (kind 3) keeps a fast rank from overwriting a word a slow one has not read yet.
__global__ void __launch_bounds__(64) k_xchg_selftest(tg_part p, int d, int rounds, int32_t* result) {
  const int q = threadIdx.x;
  const int w4 = d / 4;
  int bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    const int64_t par = r & 1;
    if (q < p.world) {
      float4* slot = reinterpret_cast<float4*>(p.push_in[q]) + ((par * p.world + p.rank) * p.push_max) * w4;
      st_sys(slot, make_float4(__int_as_float((p.rank << 16) | r), 0.f, 0.f, 0.f));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (q < p.world)
      __hip_atomic_store(p.flags[q] + 2 * TG_MAX_RANKS + p.rank, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (q < p.world) {
      if (!wait_flag(p.flags[p.rank] + 2 * TG_MAX_RANKS + q, (uint32_t)r)) bad |= 1;
      const float4* slot = reinterpret_cast<const float4*>(p.push_in[p.rank]) + ((par * p.world + q) * p.push_max) * w4;
      if (__float_as_int(ld_sys(slot).x) != ((q << 16) | r)) bad |= 2;
      __hip_atomic_store(p.flags[q] + 3 * TG_MAX_RANKS + p.rank, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (!wait_flag(p.flags[p.rank] + 3 * TG_MAX_RANKS + q, (uint32_t)r)) bad |= 4;
    }
  }
  if (bad) atomicOr(reinterpret_cast<unsigned*>(result), (unsigned)bad);
}

}  // namespace tg

using namespace tg;

extern "C" int tg_xchg_selftest(const tg_part* p, int32_t d, int32_t rounds, int32_t* result_dev, void* stream) {
  if (!p || !result_dev || p->world < 1 || p->world > TG_MAX_RANKS || rounds < 1 || d < 4 || (d % 4) || p->push_max < 1) return TG_EINVAL;
  hipLaunchKernelGGL(k_xchg_selftest, dim3(1), dim3(64), 0, as_stream(stream), *p, (int)d, (int)rounds, result_dev);
  return check_launch("tg_xchg_selftest");
}

extern "C" int tg_part_step(const tg_model* m, const tg_tcsr* g, const tg_step_io* io, const tg_part* p, void* ws,
                            size_t ws_bytes, void* aws, size_t aws_bytes, void* stream) {
  if (!m || !g || !io || !p || !io->embed_only || !io->h || !m->pending_vals) return TG_EINVAL;
  if (p->world < 1 || p->world > TG_MAX_RANKS || p->rank < 0 || p->rank >= p->world || !p->step_dev || !p->cur_step ||
      !p->ticket || !p->err || p->Bg <= 0 || (m->d % 4))
    return TG_EINVAL;
  if (m->row_of && !p->row_of) return TG_EINVAL;
  hipStream_t st = as_stream(stream);
  const int d = m->d;
  const int64_t pull_rows = std::max<int64_t>(p->serve_cap, (int64_t)p->world * p->pull_max);
  const unsigned g1 = (unsigned)std::min<int64_t>(256, std::max<int64_t>(8, cdiv(std::max<int64_t>(pull_rows * (d / 4 + 1), 2 * p->Bg), 256)));
  hipLaunchKernelGGL(k_part_begin, dim3(g1), dim3(256), 0, st, *m, *p);
  // STEP 4-6 for this rank's own winners (planned, owner-filtered: tg_stream_writeback's kernels on the staged slices).
  // The first launch (STEP 4 + 5, or STEP 4 alone with msg_src = right) reads no h(t-) and nothing the attention block
  // writes: it rides on the embedding step's fc1 / fc2 launch where that launch hosts riders (else it shares the launch
  // of the PUSH); the second waits for the pushed rows
  WritebackArgs wa{};
  wa.B = p->Bg; wa.src = p->st_src; wa.dst = p->st_dst; wa.eids = p->st_eids; wa.upos = p->st_mine_node; wa.index = p->st_mine_index;
  wa.ts = p->st_ts32; wa.n_upos = p->st_n_mine; wa.err = io->err;
  wa.rows = io->h; wa.left_row = p->st_left_row; wa.owner = p->owner; wa.my_rank = p->rank; wa.new_from_pending = 1;
  static const int ride_knob = getenv("TG_PART_WB_RIDER") ? atoi(getenv("TG_PART_WB_RIDER")) : 1;  // tuning knob: 0 = own launch
  WbRider wr{};
  wr.m = *m; wr.a = wa; wr.planned0 = 1;
  bool rode = false;
  int rc;
  if ((rc = stream_step_ext(m, g, io, ws, ws_bytes, st, ride_knob ? &wr : nullptr, &rode)) != TG_OK) return rc;
  if ((rc = part_push_wb0_launch(m, wa, p, io->h, !rode, st)) != TG_OK) return rc;
  if ((rc = part_wb1_launch(m, wa, p, 3 * io->B, st)) != TG_OK) return rc;
  // the eager updater for the nodes that have just received a message: pending[row] = updater(upd memory, mailbox)
  if ((rc = apply_messages_rows(m, p->st_mine_row, p->st_mine32, p->st_n_mine, p->mine_cap, io->err, aws, aws_bytes, st)) != TG_OK)
    return rc;
  return check_launch("tg_part_step");
}

// ---- windows: memory of this rank that its peers' kernels store into
extern "C" int tg_xchg_alloc(size_t bytes, void** out) {
  if (!out || !bytes) return TG_EINVAL;
  void* p = nullptr;
  // fine-grained: peers' stores must be seen by a kernel that is already running here
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) {
    set_hip_error(e, "tg_xchg_alloc");
    return TG_EHIP;
  }
  if ((e = hipMemset(p, 0, bytes)) != hipSuccess) {
    set_hip_error(e, "tg_xchg_alloc memset");
    return TG_EHIP;
  }
  *out = p;
  return TG_OK;
}
extern "C" int tg_xchg_clear(void* p, size_t bytes) {
  if (!p) return TG_EINVAL;
  hipError_t e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    set_hip_error(e, "tg_xchg_clear");
    return TG_EHIP;
  }
  return TG_OK;
}
extern "C" int tg_xchg_free(void* p) {
  if (p && hipFree(p) != hipSuccess) return TG_EHIP;
  return TG_OK;
}
extern "C" int tg_ipc_export(void* p, uint8_t* handle64) {
  if (!p || !handle64) return TG_EINVAL;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_ipc_export");
    return TG_EHIP;
  }
  memcpy(handle64, &h, 64);
  return TG_OK;
}
extern "C" int tg_ipc_import(const uint8_t* handle64, void** out) {
  if (!handle64 || !out) return TG_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  hipError_t e = hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    set_hip_error(e, "tg_ipc_import");
    return TG_EHIP;
  }
  return TG_OK;
}
extern "C" int tg_ipc_close(void* p) {
  if (p && hipIpcCloseMemHandle(p) != hipSuccess) return TG_EHIP;
  return TG_OK;
}
