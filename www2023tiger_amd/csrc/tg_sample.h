// Temporal-neighbour search helpers and the fused sampler of tg_stream_step as device functions, so that the collate part of
// a step (sampler + centres) can also RIDE on another launch (CollateRider, below).  Reference: tiger/data/graph.py:44-127,
// tiger/data/data_loader.py:77-131.  Integer work: bit-exact.
#pragma once
#include "tg_common.h"

namespace tg {

// number of entries of node `nid` with ts < t  (np.searchsorted(..., side='left'), graph.py:51)
__device__ __forceinline__ int64_t prefix_end(const tg_tcsr& g, int64_t nid, double t, int64_t* start) {
  if (nid < 0 || nid >= g.num_node) {
    *start = 0;
    return 0;
  }
  int64_t lo = g.indptr[nid], hi = g.indptr[nid + 1];
  *start = lo;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (g.ts[mid] < t)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// The same count, searched by the G lanes of a query group together: every round the lanes probe G
// evenly spaced positions of the remaining interval and the group's ballot bits (timestamps are sorted,
// so "ts[p] < t" is a run of ones followed by zeros) pick the sub-interval - log_{G+1}(deg) dependent
// memory round trips instead of log_2(deg); for a popular item with thousands of events that is 3 instead
// of 12, and the longest search sets the duration of the sampling kernel.  All lanes of the wavefront run
// the loop together (the ballot is a wavefront operation); finished groups idle.
template <int G>
__device__ __forceinline__ int64_t prefix_end_group(const tg_tcsr& g, int64_t nid, double t, int64_t* start, int sub) {
  int64_t lo = 0, hi = 0;
  if (nid >= 0 && nid < g.num_node) {
    lo = g.indptr[nid];
    hi = g.indptr[nid + 1];
  }
  *start = lo;
  const int shift = G == 64 ? 0 : (lane_id() / G) * G;
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  while (__any(lo < hi)) {
    const int64_t n = hi - lo;
    const int64_t pos = lo + ((int64_t)(sub + 1) * n) / (G + 1);  // in [lo, hi) when n > 0
    const bool pred = n > 0 && g.ts[pos] < t;
    const int c = __popcll((__ballot(pred) >> shift) & gmask);  // probes 0 .. c-1 are below t
    if (n > 0) {
      const int64_t below = lo + ((int64_t)c * n) / (G + 1);        // probe c-1 (for c > 0)
      const int64_t above = lo + ((int64_t)(c + 1) * n) / (G + 1);  // probe c   (for c < G)
      if (c < G) hi = above;
      if (c > 0) lo = below + 1;
    }
  }
  return lo;
}

// The sampler of the fused step: query q of cat[src, dst, neg] is built on the fly from the batch arrays (at stream offset
// `o`), written out for the later stages (ids, float32 times, edge ids), its K most recent edges before t (strict '<',
// graph.py:94-127) are copied by G lanes per query.  Workgroup `bid` of `nblk` (256 threads each).
struct SampleBatchArgs {
  tg_tcsr g;
  int64_t B;
  const int64_t *src, *dst, *neg;
  const double* ts;
  const int64_t* eids;
  const int64_t* off;  // nullable: device-resident stream offset
  int K;
  int64_t* nids3;
  float* ts3f;
  int64_t* eids_b;
  int64_t *o_nbr, *o_eid;
  float* o_ts;
  uint8_t* mark;       // nullable: involved byte flags
  uint32_t* tmin_key;  // nullable (lazy restart): complemented order-preserving key of the batch's earliest time
};
template <int G>
__device__ __forceinline__ void sample_batch_body(const SampleBatchArgs& a, int64_t o, unsigned bid, unsigned nblk) {
  constexpr int GPB = 256 / G;
  const tg_tcsr& g = a.g;
  const int sub = threadIdx.x % G;
  const int64_t B = a.B, Q = 3 * B;
  const int K = a.K;
  float tmin = INFINITY;  // earliest event time of the batch in float32 (`ts.min()` of train_self_supervised.py:162)
  for (int64_t q = (int64_t)bid * GPB + threadIdx.x / G; q < Q; q += (int64_t)nblk * GPB) {
    const int64_t e = q % B;
    const int r = (int)(q / B);
    const int64_t nid = r == 0 ? a.src[o + e] : (r == 1 ? a.dst[o + e] : a.neg[o + e]);
    const double t = a.ts[o + e];
    if (r == 0) tmin = fminf(tmin, (float)t);
    if (sub == 0) {
      a.nids3[q] = nid;
      a.ts3f[q] = (float)t;
      if (r == 0) a.eids_b[e] = a.eids[o + e];
    }
    int64_t start;
    const int64_t end = prefix_end_group<G>(g, nid, t, &start, sub);
    for (int j = sub; j < K; j += G) {
      const int64_t p = end - K + j;
      int64_t nb = 0, ed = 0;
      float tt = 0.f;
      if (p >= start) {
        nb = g.nbr[p];
        ed = (int64_t)((uint32_t)g.eid[p] & 0x7fffffffu);
        tt = (float)g.ts[p];
      }
      const int64_t w = q * K + j;
      a.o_nbr[w] = nb;
      a.o_eid[w] = ed;
      a.o_ts[w] = tt;
      if (a.mark) a.mark[nb] = 1;
    }
    if (a.mark && sub == 0 && nid >= 0 && nid < g.num_node) a.mark[nid] = 1;
  }
  if (a.tmin_key) {  // lazy restart only: one atomic per block on the complemented order-preserving key (slot starts at 0)
    __shared__ float s_tmin[4];
    for (int sh = 32; sh > 0; sh >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, sh, TG_WAVE));
    if (lane_id() == 0) s_tmin[threadIdx.x >> 6] = tmin;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float v = fminf(fminf(s_tmin[0], s_tmin[1]), fminf(s_tmin[2], s_tmin[3]));
      if (v < INFINITY) atomicMax(a.tmin_key, ~(uint32_t)orderable(v));
    }
  }
}

// Collate rider (tg_step_io.prefetch_state): sampler and centres of the NEXT batch of a resident stream as the first
// `blocks` workgroups of the step's last launch (the product that refreshes the query rows of the batch's positive nodes).
// The sampler reads the graph only; the centres read state that is final once the updater has run - and nothing in that
// last product's launch reads what they write (step workspace: query arrays, neighbour lists, centre rows, snapshot,
// dedup slots).  The next step then starts with its attention core.  256-thread workgroups, K <= 16.
struct CollateRider {
  SampleBatchArgs s;   // (mark, tmin_key: null - lean steps only)
  CentresRider cr;     // cr.blocks: the centres' workgroups
  int64_t stream_len;  // the batch [*off, *off + B) must lie inside the stream, else the rider does nothing
  unsigned sblocks;    // the sampler's workgroups
  unsigned blocks;     // sblocks + cr.blocks rounded up to a multiple of 8 (the host launch's XCD map stays)
  unsigned last;       // riders behind the host launch's own blocks (else in front); set by the launcher
  // which halves ride on THIS launch (0 = both): 1 = the sampler alone - it reads the graph and the stream only, so it can
  // share an earlier launch than the centres (fc2's, once the write-back rider on fc1's launch has advanced the offset);
  // 2 = the centres alone (the sampler rode elsewhere)
  unsigned parts;
  __device__ __forceinline__ void run(unsigned bid) const {
    const int64_t o = *s.off;
    if (o + s.B > stream_len) return;
    if (bid < sblocks) {
      sample_batch_body<16>(s, o, bid, sblocks);
    } else if (bid - sblocks < cr.blocks) {
      centres_direct_body(cr.m, 3 * s.B, RawIds{s.src, s.dst, s.neg, s.ts, o, s.B}, cr.nf, cr.out, cr.da, cr.pos,
                          (int64_t)(bid - sblocks) * blockDim.x + threadIdx.x, (int64_t)cr.blocks * blockDim.x);
    }
  }
};

}  // namespace tg
