// Device helpers of the partitioned multi-GPU step (tg_part.hip, tg_memory.hip): hand-offs between RANKS through windows.
// Every byte a peer's kernel stores into a window and every flag is written and read with SYSTEM-scope accesses (sc0 sc1:
// write-through stores, loads that miss every cache of the reading GPU), so no fence is needed on either side: a storing
// wave waits for its stores (vmcnt(0)), the workgroup's barrier, one lane counts the workgroup in, the LAST workgroup
// raises this rank's flag at every peer; a consuming workgroup polls its own flags (one lane, bounded), its barrier, then
// system-scope loads of the rows.  (Guide: MI355X_MICROARCH.md, visibility - the "{sc0 sc1 stores and loads both sides}"
// form, widened from agent to system scope because producer and consumer sit on different GPUs.)
#pragma once
#include "tg_common.h"

namespace tg {

// bounded wait for `flag >= epoch` (one lane).  The peer's kernel that raises the flag precedes, in that peer's stream,
// every wait of that peer for this step, so the wait ends unless a peer died: then the timeout bit is raised and the step
// goes on (wrong rows, no hang; the caller reads the error word).  The timeout is STICKY: a wait entered with the bit
// already set does not spin at all, so a dead peer costs ONE bound per run - not two per remaining step -
// and the host, which reads the word after every replay (dist.py), stops the run there.
__device__ __forceinline__ bool wait_flag(const uint32_t* flag, uint32_t epoch, const uint32_t* err = nullptr) {
  if (err && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & TG_ERR_XCHG_TIMEOUT)) return false;
  for (unsigned spin = 0; spin < (1u << 22); ++spin) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= epoch) return true;
    __builtin_amdgcn_s_sleep(8);
  }
  return false;
}
// every workgroup `bid` of a consuming kernel, before it loads (ld_sys) what the peers stored for `kind`.  ONE lane of the
// kernel (workgroup 0's) polls the flags the peers raise - system-scope loads that go to memory every time: a thousand
// workgroups polling them (four ranks rehearsed on one GPU) starve the very kernels they wait for - and opens a gate
// word of this GPU (p.ticket[2 + kind], agent scope) that the other workgroups poll in the L2
__device__ __forceinline__ void wait_peers(const tg_part& p, int kind, uint32_t epoch, unsigned bid) {
  if (threadIdx.x == 0) {
    uint32_t* gate = p.ticket + 2 + kind;
    if (bid == 0) {
      const uint32_t* mine = p.flags[p.rank] + (size_t)kind * TG_MAX_RANKS;
      bool ok = true;
      for (int q = 0; q < p.world; ++q) ok = wait_flag(mine + q, epoch, p.err) && ok;
      if (!ok) atomicOr(p.err, TG_ERR_XCHG_TIMEOUT);
      __hip_atomic_store(gate, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (unsigned spin = 0; spin < (1u << 24); ++spin) {  // (workgroup 0 gives up first and opens the gate anyway)
        if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch) break;
        if ((spin & 1023u) == 1023u && (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & TG_ERR_XCHG_TIMEOUT)) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  __syncthreads();
}
// every workgroup `bid` of the `nblk` workgroups of a producing kernel, after its stores (st_sys)
__device__ __forceinline__ void signal_peers(const tg_part& p, int kind, uint32_t epoch, unsigned nblk) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(p.ticket + kind, 1u);
    if (prev == nblk - 1) {  // every workgroup's stores are complete: raise this rank's flag at every peer
      p.ticket[kind] = 0u;
      for (int q = 0; q < p.world; ++q)
        __hip_atomic_store(p.flags[q] + (size_t)kind * TG_MAX_RANKS + p.rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// launches of tg_part_step that live beside the write-back kernels (tg_memory.hip):
// PUSH (h(t-) of winning positions of nodes owned elsewhere -> their owners' windows, signal) together with the first
// write-back launch of the owner's own winners (STEP 4 + 5, or STEP 4 alone with msg_src = right: neither reads h(t-))
int part_push_wb0_launch(const tg_model* m, const WritebackArgs& a, const tg_part* p, const float* h, bool with_wb0,
                         hipStream_t st);
// the second write-back launch (STEP 6, or STEP 5 + 6) behind the wait for the pushed rows, which it reads in the window
// (rows >= hi_from of a.left_row); ends the step: advances the step counter
int part_wb1_launch(const tg_model* m, const WritebackArgs& a, const tg_part* p, int64_t hi_from, hipStream_t st);

}  // namespace tg
