// Temporal CSR + temporal neighbour sampling kernels.
// Reference: tiger/data/graph.py:11-155,226-241; tiger/data/data_loader.py:61-75;
// tiger/model/utils.py:19-27.  Integer work: results are bit-exact.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "tg_common.h"
#include "tg_sample.h"

namespace tg {

static thread_local std::string g_hip_error;
void set_hip_error(hipError_t e, const char* what) {
  g_hip_error = std::string(what) + ": " + hipGetErrorString(e);
}

}  // namespace tg

namespace tg {
__device__ float4 g_zero16_store = {0.f, 0.f, 0.f, 0.f};
const float* zero_line() {
  // a device symbol has one address PER DEVICE: cached per device (a process may drive several GPUs)
  constexpr int MAXD = 64;
  static const float* cache[MAXD] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXD) dev = -1;
  if (dev >= 0 && cache[dev]) return cache[dev];
  void* a = nullptr;
  if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_zero16_store)) != hipSuccess) return nullptr;
  if (dev >= 0) cache[dev] = (const float*)a;
  return (const float*)a;
}
}  // namespace tg

extern "C" int tg_abi_version(void) { return TG_ABI_VERSION; }
extern "C" const char* tg_last_hip_error(void) { return tg::g_hip_error.c_str(); }

// ---------------------------------------------------------------------------------
// Host-side T-CSR build (initialisation, graph.py:30-36 + data2adjlist :226-241).
// Entry 2i is event i seen from src (flag 0), entry 2i+1 is event i seen from dst
// (flag 1); a counting sort by owner keeps that order, then each node's run is
// stably sorted by time if it is not already ascending.
// ---------------------------------------------------------------------------------
extern "C" int tg_tcsr_build_host(int64_t E, const int64_t* src, const int64_t* dst, const double* ts,
                                  const int64_t* eid, int64_t num_node, int64_t* indptr, double* ts_out,
                                  int32_t* nbr_out, int32_t* eid_out) {
  if (E < 0 || num_node <= 0 || num_node > 0x7fffffffLL) return TG_EINVAL;
  for (int64_t i = 0; i < E; ++i) {
    if (src[i] < 0 || src[i] >= num_node || dst[i] < 0 || dst[i] >= num_node) return TG_EINVAL;
    if (eid[i] < 0 || eid[i] > 0x7fffffffLL) return TG_EINVAL;
  }
  std::fill(indptr, indptr + num_node + 1, (int64_t)0);
  for (int64_t i = 0; i < E; ++i) {
    indptr[src[i] + 1]++;
    indptr[dst[i] + 1]++;
  }
  for (int64_t n = 0; n < num_node; ++n) indptr[n + 1] += indptr[n];
  std::vector<int64_t> cur(indptr, indptr + num_node);
  for (int64_t i = 0; i < E; ++i) {
    int64_t p = cur[src[i]]++;
    ts_out[p] = ts[i];
    nbr_out[p] = (int32_t)dst[i];
    eid_out[p] = (int32_t)eid[i];
    p = cur[dst[i]]++;
    ts_out[p] = ts[i];
    nbr_out[p] = (int32_t)src[i];
    eid_out[p] = (int32_t)((uint32_t)eid[i] | 0x80000000u);
  }
  std::vector<int64_t> perm;
  std::vector<double> t_tmp;
  std::vector<int32_t> a_tmp, b_tmp;
  for (int64_t n = 0; n < num_node; ++n) {
    const int64_t lo = indptr[n], hi = indptr[n + 1];
    bool sorted = true;
    for (int64_t p = lo + 1; p < hi; ++p)
      if (ts_out[p] < ts_out[p - 1]) {
        sorted = false;
        break;
      }
    if (sorted) continue;
    const int64_t len = hi - lo;
    perm.resize(len);
    std::iota(perm.begin(), perm.end(), (int64_t)0);
    std::stable_sort(perm.begin(), perm.end(),
                     [&](int64_t a, int64_t b) { return ts_out[lo + a] < ts_out[lo + b]; });
    t_tmp.assign(ts_out + lo, ts_out + hi);
    a_tmp.assign(nbr_out + lo, nbr_out + hi);
    b_tmp.assign(eid_out + lo, eid_out + hi);
    for (int64_t k = 0; k < len; ++k) {
      ts_out[lo + k] = t_tmp[perm[k]];
      nbr_out[lo + k] = a_tmp[perm[k]];
      eid_out[lo + k] = b_tmp[perm[k]];
    }
  }
  return TG_OK;
}

namespace tg {

// G lanes cooperate on one query: they search the prefix end together (above), then copy
// the K-entry tail with one lane per slot.
template <int G>
__global__ void __launch_bounds__(256) k_sample_recent_edges(tg_tcsr g, int64_t Q, const int64_t* __restrict__ nids,
                                                             const double* __restrict__ qts, int K,
                                                             int64_t* __restrict__ o_nbr, int64_t* __restrict__ o_eid,
                                                             float* __restrict__ o_ts, int64_t* __restrict__ o_dir,
                                                             uint8_t* __restrict__ mark,
                                                             const float* __restrict__ qts32 = nullptr) {
  constexpr int GPB = 256 / G;
  const int sub = threadIdx.x % G;
  for (int64_t q = (int64_t)blockIdx.x * GPB + threadIdx.x / G; q < Q; q += (int64_t)gridDim.x * GPB) {
    const int64_t nid = nids[q];
    int64_t start;
    const int64_t end = prefix_end_group<G>(g, nid, qts ? qts[q] : (double)qts32[q], &start, sub);
    for (int j = sub; j < K; j += G) {
      const int64_t p = end - K + j;
      int64_t nb = 0, ed = 0, dr = 0;
      float tt = 0.f;
      if (p >= start) {
        nb = g.nbr[p];
        const uint32_t e = (uint32_t)g.eid[p];
        ed = (int64_t)(e & 0x7fffffffu);
        dr = (int64_t)(e >> 31);
        tt = (float)g.ts[p];
      }
      const int64_t o = q * K + j;
      o_nbr[o] = nb;
      o_eid[o] = ed;
      o_ts[o] = tt;
      if (o_dir) o_dir[o] = dr;
      if (mark) mark[nb] = 1;  // byte flag, plain store: every writer stores the same value, no atomics
    }
    if (mark && sub == 0 && nid >= 0 && nid < g.num_node) mark[nid] = 1;
  }
}

// Fused form used by tg_stream_step (body: tg_sample.h, sample_batch_body): query q of cat[src,dst,neg] is built on the
// fly from the batch arrays (optionally at a device-resident stream offset) and also written out for the later stages
// (ids, float32 times, edge ids); the centres of a lean step ride on the launch as extra workgroups at its end.
template <int G>
__global__ void __launch_bounds__(256) k_sample_batch(SampleBatchArgs a, CentresRider cr) {
  const unsigned sblocks = gridDim.x - cr.blocks;  // the sampler's share of the grid
  const int64_t o = a.off ? *a.off : 0;
  if (blockIdx.x >= sblocks) {  // rider: the attention centres of a lean step (tg_common.h)
    centres_direct_body(cr.m, 3 * a.B, RawIds{a.src, a.dst, a.neg, a.ts, o, a.B}, cr.nf, cr.out, cr.da, cr.pos,
                        (int64_t)(blockIdx.x - sblocks) * blockDim.x + threadIdx.x, (int64_t)cr.blocks * blockDim.x);
    return;
  }
  sample_batch_body<G>(a, o, blockIdx.x, sblocks);
}

// ---- lazy restart with the static restarter (train_self_supervised.py:152-163, tiger.py:594-609,
// restarters.py:262-277); contract in tiger_hip.h (tg_lazy_restart).  One wavefront per bitmap word: lane l
// decides for node 64 w + l (involved this batch and not yet up to date), needing lanes search the time of
// their node's last event before the batch's earliest time (float32 query, as the reference's history call),
// then the wavefront copies the surrogate rows of each needing node.  The word's has-message and uptodate
// bits have a single writer (this wavefront): plain stores.
// rlist / rlist32 (nullable; static form): the re-initialised nodes are also listed (compacted through *n_restarted) and
// their rows of the per-node centre-row table (tg_model.c_table) rewritten - c_v = right_static[v] + nfeat[v], the node
// has no pending message any more - so that the step can refresh their query rows (tg_model.g_table) right behind
__global__ void __launch_bounds__(256) k_lazy_restart(tg_tcsr g, tg_model m, tg_lazy_restart lz,
                                                      const uint8_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ tmin_key,
                                                      int32_t* __restrict__ n_restarted, int64_t* __restrict__ rlist,
                                                      int32_t* __restrict__ rlist32) {
  const int lane = lane_id();
  const int64_t W = (m.n_nodes + 63) / 64;
  const int64_t b = lz.batch_dev ? *lz.batch_dev : 0;
  const bool trig = b >= 0 && b < lz.n_trigger && lz.trigger[b] != 0;
  if (!trig && *lz.restarting_dev == 0) return;
  uint32_t key = ~*tmin_key;  // inverse of orderable(float)
  key = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key;
  const double t = (double)__uint_as_float(key);
  const int w4 = m.d / 4;
  const float4* sl = reinterpret_cast<const float4*>(lz.static_left);
  const float4* sr = reinterpret_cast<const float4*>(lz.static_right);
  float4* left = reinterpret_cast<float4*>(m.left_vals);
  float4* right = reinterpret_cast<float4*>(m.right_vals);
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < W; w += (int64_t)gridDim.x * 4) {
    const bool keep_msg = lz.list && lz.keep_msg_bits && !trig;  // (the caller's restart clears the listed nodes' bits)
    const uint64_t upd = trig ? 0ull : lz.uptodate[w];  // a trigger forgets who is up to date ...
    const uint64_t msg = (trig || keep_msg) ? 0ull : m.has_msg[w];  // ... and drops every pending message (msg_store.clear())
    const int64_t node = w * 64 + lane;
    const bool need_l = node < m.n_nodes && flags[node] != 0 && !((upd >> lane) & 1ull);
    const unsigned long long need = __ballot(need_l);
    if (lz.list) {  // list form: the caller's restarter re-initialises these nodes; only the bookkeeping happens here
      int base = 0;
      if (lane == 0 && need) base = atomicAdd(n_restarted, __popcll(need));
      base = __shfl(base, 0, TG_WAVE);
      if (need_l) lz.list[base + __popcll(need & ((1ull << lane) - 1ull))] = node;
      if (lane == 0 && (trig || need)) {
        lz.uptodate[w] = upd | need;
        if (!keep_msg) m.has_msg[w] = msg & ~need;
      }
      continue;
    }
    float pt_l = 0.f;
    if (need_l) {
      int64_t start;
      const int64_t end = prefix_end(g, node, t, &start);
      if (end > start) pt_l = (float)g.ts[end - 1];
    }
    int lbase = 0;
    if (rlist && need) {
      if (lane == 0) lbase = atomicAdd(n_restarted, __popcll(need));
      lbase = __shfl(lbase, 0, TG_WAVE);
      if (need_l) {
        const int at = lbase + __popcll(need & ((1ull << lane) - 1ull));
        rlist[at] = node;
        rlist32[at] = (int32_t)node;
      }
    }
    float4* ctab = reinterpret_cast<float4*>(m.c_table);
    const float4* nf4 = reinterpret_cast<const float4*>(m.nfeats);
    for (unsigned long long todo = need; todo; todo &= todo - 1) {
      const int k = __ffsll(todo) - 1;
      const int64_t v = w * 64 + k;
      const float pt = __shfl(pt_l, k, TG_WAVE);
      for (int c = lane; c < w4; c += TG_WAVE) {
        const float4 rv = sr[v * w4 + c];
        left[v * w4 + c] = sl[v * w4 + c];
        right[v * w4 + c] = rv;
        if (rlist && ctab) {
          float4 cv = rv;
          if (nf4) { const float4 f = nf4[v * w4 + c]; cv.x += f.x; cv.y += f.y; cv.z += f.z; cv.w += f.w; }
          ctab[v * w4 + c] = cv;
        }
      }
      if (lane == 0) {
        m.left_ts[v] = pt;
        m.right_ts[v] = pt;
        if (m.left_active) m.left_active[v] = 1;
        if (m.right_active) m.right_active[v] = 1;
      }
    }
    if (lane == 0 && (trig || need)) {
      lz.uptodate[w] = upd | need;
      m.has_msg[w] = msg & ~need;
      if (need && !rlist) atomicAdd(n_restarted, __popcll(need));
    }
  }
  if (trig && blockIdx.x == 0 && threadIdx.x == 0) *lz.restarting_dev = 1;
  if (lz.tmin && blockIdx.x == 0 && threadIdx.x == 0) *lz.tmin = (float)t;
}

int lazy_restart_launch(const tg_tcsr* g, const tg_model* m, const tg_lazy_restart* lz, const uint8_t* flags,
                        const uint32_t* tmin_key, int32_t* n_restarted, hipStream_t st, int64_t* rlist, int32_t* rlist32) {
  if (!lz->trigger || !lz->restarting_dev || !lz->uptodate) return TG_EINVAL;
  if (lz->list ? (lz->static_left || lz->static_right) : (!lz->static_left || !lz->static_right)) return TG_EINVAL;
  if (lz->list && rlist) return TG_EINVAL;
  const int64_t W = (m->n_nodes + 63) / 64;
  hipLaunchKernelGGL(k_lazy_restart, dim3(flat_grid(W, 4)), dim3(256), 0, st, *g, *m, *lz, flags, tmin_key, n_restarted,
                     rlist, rlist32);
  return check_launch("lazy_restart");
}

int sample_batch_launch(const tg_tcsr* g, int64_t B, const int64_t* src, const int64_t* dst, const int64_t* neg,
                        const double* ts, const int64_t* eids, const int64_t* off, int32_t K, int64_t* nids3,
                        float* ts3f, int64_t* eids_b, int64_t* o_nbr, int64_t* o_eid, float* o_ts, uint8_t* mark,
                        hipStream_t st, uint32_t* tmin_key, const CentresRider* rider) {
  const int64_t Q = 3 * B;
  const CentresRider cr = rider ? *rider : CentresRider{};
  const SampleBatchArgs a{*g, B, src, dst, neg, ts, eids, off, K, nids3, ts3f, eids_b, o_nbr, o_eid, o_ts, mark, tmin_key};
  if (K <= 16)
    TG_KLAUNCH(k_sample_batch<16>, dim3(flat_grid(Q, 16) + cr.blocks), dim3(256), 0, st, a, cr);
  else
    TG_KLAUNCH(k_sample_batch<64>, dim3(flat_grid(Q, 4) + cr.blocks), dim3(256), 0, st, a, cr);
  return check_launch("sample_batch");
}

// One wavefront per query.  Walk the prefix backwards 64 entries at a time; an entry
// is kept if no more recent entry (this chunk or earlier chunks) has the same
// neighbour.  The j-th kept entry (j = 0 most recent) lands in output slot K-1-j.
__global__ void __launch_bounds__(256) k_sample_recent_nodes(tg_tcsr g, int64_t Q, const int64_t* __restrict__ nids,
                                                             const double* __restrict__ qts, int K,
                                                             int64_t* __restrict__ o_nbr, int64_t* __restrict__ o_eid,
                                                             float* __restrict__ o_ts, int64_t* __restrict__ o_dir,
                                                             const float* __restrict__ qts32 = nullptr) {
  __shared__ int s_new[4][TG_WAVE];
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  for (int64_t q = (int64_t)blockIdx.x * 4 + wv; q < Q; q += (int64_t)gridDim.x * 4) {
    int64_t start;
    const int64_t end = prefix_end(g, nids[q], qts ? qts[q] : (double)qts32[q], &start);  // (second hop: float32 query times)
    int c = 0;        // wave-uniform: number collected so far
    int mycol = -1;   // lane j holds the neighbour id of the j-th collected entry
    for (int64_t chunk_end = end; chunk_end > start && c < K; chunk_end -= TG_WAVE) {
      const int64_t p = chunk_end - 1 - lane;
      const bool valid = p >= start;
      const int v = valid ? g.nbr[p] : -1;
      bool isnew = valid;
      for (int i = 0; i < TG_WAVE; ++i) {
        const int vi = __shfl(v, i, TG_WAVE);
        if (i < lane && vi == v) isnew = false;
      }
      for (int j = 0; j < c; ++j) {
        const int cj = __shfl(mycol, j, TG_WAVE);
        if (cj == v) isnew = false;
      }
      const unsigned long long m = __ballot(isnew);
      const int slot = c + __popcll(m & ((1ull << lane) - 1ull));
      if (isnew && slot < K) {
        const uint32_t e = (uint32_t)g.eid[p];
        const int64_t o = q * K + (K - 1 - slot);
        o_nbr[o] = v;
        o_eid[o] = (int64_t)(e & 0x7fffffffu);
        o_ts[o] = (float)g.ts[p];
        if (o_dir) o_dir[o] = (int64_t)(e >> 31);
        if (slot < TG_WAVE) s_new[wv][slot] = v;
      }
      const int nnew = __popcll(m);
      __builtin_amdgcn_wave_barrier();
      if (lane >= c && lane < c + nnew && lane < K) mycol = s_new[wv][lane];
      __builtin_amdgcn_wave_barrier();
      c += nnew;
    }
    if (c > K) c = K;
    for (int j = lane; j < K - c; j += TG_WAVE) {  // left padding
      const int64_t o = q * K + j;
      o_nbr[o] = 0;
      o_eid[o] = 0;
      o_ts[o] = 0.f;
      if (o_dir) o_dir[o] = 0;
    }
  }
}

// ---- MT19937 exactly as numpy's legacy RandomState (randomkit) ---------------------
__device__ __forceinline__ void mt_regen(uint32_t* key, int lane) {
  // sequential dependency of distance 397/227: regenerate in three independent
  // stripes so that the wave can work in parallel inside a stripe.
  constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
  for (int base = 0; base < 227; base += TG_WAVE) {  // kk in [0,227): uses key[kk+397] (old)
    const int kk = base + lane;
    uint32_t y = 0;
    if (kk < 227) y = (key[kk] & UP) | (key[kk + 1] & LO);
    __builtin_amdgcn_wave_barrier();
    if (kk < 227) key[kk] = key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
    __builtin_amdgcn_wave_barrier();
  }
  // kk in [227,623): uses key[kk-227] (new); split so a stripe never reads what it writes
  for (int seg = 227; seg < 623; seg += 227) {
    const int seg_end = (seg + 227 < 623) ? seg + 227 : 623;
    for (int base = seg; base < seg_end; base += TG_WAVE) {
      const int kk = base + lane;
      uint32_t y = 0;
      if (kk < seg_end) y = (key[kk] & UP) | (key[kk + 1] & LO);
      __builtin_amdgcn_wave_barrier();
      if (kk < seg_end) key[kk] = key[kk - 227] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0) {
    const uint32_t y = (key[623] & UP) | (key[0] & LO);
    key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// The random stream is consumed per non-empty query in query order (graph.py:103),
// and randint's masked rejection makes the draw count data dependent, so ONE wavefront
// walks the queries.  Lane 0 draws; all lanes help regenerate the state and copy.
__global__ void __launch_bounds__(64) k_sample_uniform(tg_tcsr g, int64_t Q, const int64_t* __restrict__ nids,
                                                       const double* __restrict__ qts, int K, uint32_t* mt_state,
                                                       int64_t* __restrict__ o_nbr, int64_t* __restrict__ o_eid,
                                                       float* __restrict__ o_ts, int64_t* __restrict__ o_dir,
                                                       const float* __restrict__ qts32 = nullptr) {
  __shared__ uint32_t key[624];
  __shared__ int s_pos;
  __shared__ int64_t s_sel[TG_WAVE];
  const int lane = lane_id();
  for (int i = lane; i < 624; i += TG_WAVE) key[i] = mt_state[i];
  if (lane == 0) s_pos = (int)mt_state[624];
  __builtin_amdgcn_wave_barrier();
  for (int64_t q = 0; q < Q; ++q) {
    int64_t start;
    const int64_t end = prefix_end(g, nids[q], qts ? qts[q] : (double)qts32[q], &start);  // (second hop: float32 query times)
    const int64_t len = end - start;
    if (len == 0) {
      for (int j = lane; j < K; j += TG_WAVE) {
        const int64_t o = q * K + j;
        o_nbr[o] = 0;
        o_eid[o] = 0;
        o_ts[o] = 0.f;
        if (o_dir) o_dir[o] = 0;
      }
      continue;
    }
    // numpy _rand_int64 / _bounded_uint64 with use_masked: rng = len-1; rng == 0 draws
    // nothing; rng <= 0xFFFFFFFF uses 32-bit draws masked to the next power of two - 1.
    const uint64_t rng = (uint64_t)(len - 1);
    uint32_t mask = (uint32_t)rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    if (rng == 0) {  // one entry: randint draws nothing
      if (lane < K) s_sel[lane] = start;
    } else {
      // The next words of the stream are judged 64 at a time: word j is accepted iff (tempered & mask) <= rng, the query
      // takes its first K accepted words in order and consumes the stream up to the K-th (the words behind it in the
      // window stay for the next query) - the same draws, in the same order, as K calls of the scalar rejection loop
      int got = 0;
      while (got < K) {  // wave-uniform
        int pos = s_pos;
        __builtin_amdgcn_wave_barrier();
        if (pos == 624) {
          mt_regen(key, lane);
          pos = 0;
          __builtin_amdgcn_wave_barrier();
        }
        const int avail = min(TG_WAVE, 624 - pos);
        const uint32_t v = lane < avail ? (mt_temper(key[pos + lane]) & mask) : 0xffffffffu;
        const bool acc = lane < avail && v <= (uint32_t)rng;
        const unsigned long long bal = __ballot(acc);
        const int r = __popcll(bal & ((1ull << lane) - 1ull));  // accepted words in front of this lane
        const int need = K - got;
        if (acc && r < need) s_sel[got + r] = start + (int64_t)v;
        int used = avail;
        if (__popcll(bal) >= need) {  // the need-th accepted word ends the query
          const unsigned long long last = __ballot(acc && r == need - 1);
          used = (int)__ffsll((long long)last);  // its lane + 1
          got = K;
        } else {
          got += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) s_pos = pos + used;
        __builtin_amdgcn_wave_barrier();
      }
    }
    __builtin_amdgcn_wave_barrier();
    // stable insertion sort by time (numpy's argsort on <= 16 elements is insertion sort)
    if (lane == 0) {
      for (int a = 1; a < K; ++a) {
        const int64_t x = s_sel[a];
        const double tx = g.ts[x];
        int b = a - 1;
        while (b >= 0 && g.ts[s_sel[b]] > tx) {
          s_sel[b + 1] = s_sel[b];
          --b;
        }
        s_sel[b + 1] = x;
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < K; j += TG_WAVE) {
      const int64_t p = s_sel[j];
      const uint32_t e = (uint32_t)g.eid[p];
      const int64_t o = q * K + j;
      o_nbr[o] = g.nbr[p];
      o_eid[o] = (int64_t)(e & 0x7fffffffu);
      o_ts[o] = (float)g.ts[p];
      if (o_dir) o_dir[o] = (int64_t)(e >> 31);
    }
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = lane; i < 624; i += TG_WAVE) mt_state[i] = key[i];
  if (lane == 0) mt_state[624] = (uint32_t)s_pos;
}

// ---- RandEdgeSampler.sample(1) x count on device (data_loader.py:291-294): per event one
// randint(0, n_src) then one randint(0, n_dst) of numpy's legacy RandomState - masked rejection on 32-bit
// draws, so how many words a draw consumes depends on the words.  One wavefront; the 624 tempered words of
// a state block are judged 64 at a time: word i would be accepted as a source draw (a_i) or as a destination
// draw (b_i), and whether it is asked for a source or a destination is the state of a two-state automaton
// (accepted source -> expect destination, accepted destination -> expect source, rejection -> stay).  The
// state in front of every word is an exclusive scan over the transition maps (composition of 2-bit maps,
// six shuffle steps), the event a word belongs to is a prefix popcount of the accept ballots.
__device__ __forceinline__ uint32_t bitmask_of(uint32_t rng) {
  uint32_t m = rng;
  m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
  return m;
}
__device__ __forceinline__ uint32_t map_compose(uint32_t later, uint32_t earlier) {  // x -> later(earlier(x))
  return ((later >> (earlier & 1u)) & 1u) | (((later >> ((earlier >> 1) & 1u)) & 1u) << 1);
}
__global__ void __launch_bounds__(64) k_rand_edge_pairs(uint32_t* mt_state, uint32_t rng_s, uint32_t rng_d, int64_t count,
                                                        const int64_t* __restrict__ src_list,
                                                        const int64_t* __restrict__ dst_list,
                                                        int64_t* __restrict__ out_src, int64_t* __restrict__ out_dst) {
  __shared__ uint32_t key[624];
  const int lane = lane_id();
  for (int i = lane; i < 624; i += TG_WAVE) key[i] = mt_state[i];
  int pos = (int)mt_state[624];
  __builtin_amdgcn_wave_barrier();
  const uint32_t mask_s = bitmask_of(rng_s), mask_d = bitmask_of(rng_d);
  const unsigned long long below = (1ull << lane) - 1ull;
  auto emit = [&](int64_t* out, const int64_t* list, int64_t e, uint32_t v) { out[e] = list ? list[v] : (int64_t)v; };
  if (rng_s == 0) {  // a range of one value draws nothing (numpy returns the offset without touching the state)
    for (int64_t e = lane; e < count; e += TG_WAVE) emit(out_src, src_list, e, 0u);
  }
  if (rng_d == 0) {
    for (int64_t e = lane; e < count; e += TG_WAVE) emit(out_dst, dst_list, e, 0u);
  }
  const bool both = rng_s != 0 && rng_d != 0;
  int64_t done_s = 0, done_d = 0;  // accepted source / destination draws so far (wave-uniform)
  uint32_t state = 0;              // both ranges draw: 0 = the next accepted word is a source, 1 = a destination
  const int64_t want = (rng_s == 0 && rng_d == 0) ? 0 : count;
  while ((rng_d != 0 ? done_d : done_s) < want) {
    if (pos >= 624) {
      mt_regen(key, lane);
      pos = 0;
    }
    const int i = pos + lane;
    const bool live = i < 624;
    const uint32_t w = live ? mt_temper(key[i]) : 0u;
    const uint32_t vs = w & mask_s, vd = w & mask_d;
    const bool a = live && rng_s != 0 && vs <= rng_s;
    const bool b = live && rng_d != 0 && vd <= rng_d;
    bool is_s, is_d;
    if (both) {
      uint32_t m = live ? ((a ? 1u : 0u) | ((b ? 0u : 1u) << 1)) : 2u;  // bit x = next state from state x; 2 = identity
      for (int sh = 1; sh < TG_WAVE; sh <<= 1) {  // inclusive scan of the maps
        const uint32_t prev = __shfl_up(m, sh, TG_WAVE);
        if (lane >= sh) m = map_compose(m, prev);
      }
      const uint32_t upto = __shfl_up(m, 1, TG_WAVE);
      const uint32_t st = lane == 0 ? state : ((upto >> state) & 1u);  // the state in front of this lane's word
      is_s = a && st == 0;
      is_d = b && st == 1;
      state = (__shfl(m, TG_WAVE - 1, TG_WAVE) >> state) & 1u;
    } else {
      is_s = a;
      is_d = b;
    }
    const unsigned long long ms = __ballot(is_s), md = __ballot(is_d);
    // the chunk ends the job at the word that completes the last event: the count-th accepted draw of the
    // kind that closes an event (destination when it draws, else source); later words stay unconsumed
    const unsigned long long mc = rng_d != 0 ? md : ms;
    const int64_t had = rng_d != 0 ? done_d : done_s;
    int used = min(TG_WAVE, 624 - pos);
    if (had + __popcll(mc) >= want) {  // find the lane of the (want - had)-th set bit of mc
      unsigned long long t = mc;
      for (int64_t k = 1; k < want - had; ++k) t &= t - 1;
      used = __ffsll(t);  // 1-based lane index = number of words consumed in this chunk
    }
    const unsigned long long keep = used >= 64 ? ~0ull : ((1ull << used) - 1ull);
    if (is_s && ((keep >> lane) & 1ull)) {
      const int64_t e = done_s + __popcll(ms & below);
      if (e < count) emit(out_src, src_list, e, vs);
    }
    if (is_d && ((keep >> lane) & 1ull)) {
      const int64_t e = done_d + __popcll(md & below);
      if (e < count) emit(out_dst, dst_list, e, vd);
    }
    done_s += __popcll(ms & keep);
    done_d += __popcll(md & keep);
    pos += used;
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < 624; i += TG_WAVE) mt_state[i] = key[i];
  if (lane == 0) mt_state[624] = (uint32_t)pos;
}

__global__ void k_hits(int64_t n, int K, const int64_t* __restrict__ center, const int64_t* __restrict__ nbr,
                       float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (center[i / K] == nbr[i]) ? 1.f : 0.f;
}

// one wavefront per row, lane = column (H <= 64).  id x is numbered
// 1 + #{distinct ids whose last occurrence lies to the right of x's last occurrence}.
__global__ void __launch_bounds__(256) k_anon_reindex(int64_t n, int H, const int64_t* __restrict__ in,
                                                      int64_t* __restrict__ out) {
  const int lane = lane_id();
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
    const int64_t v = lane < H ? in[r * H + lane] : -1;
    int last = lane;       // last column holding the same id
    for (int i = 0; i < H; ++i) {
      const int64_t vi = __shfl(v, i, TG_WAVE);
      if (vi == v && i > last) last = i;
    }
    const bool is_last = (lane < H) && (last == lane);
    const unsigned long long m = __ballot(is_last);
    const unsigned long long right = (last >= 63) ? 0ull : (m >> (last + 1));
    const int64_t code = 1 + __popcll(right);
    if (lane < H) out[r * H + lane] = (v == 0) ? 0 : code;
  }
}
// histories of 65 .. 128 events: two columns per lane (lane, lane + 64), the same rule
__global__ void __launch_bounds__(256) k_anon_reindex2(int64_t n, int H, const int64_t* __restrict__ in,
                                                       int64_t* __restrict__ out) {
  const int lane = lane_id();
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
    const int64_t v0 = in[r * H + lane];  // (H > 64)
    const int64_t v1 = lane + 64 < H ? in[r * H + 64 + lane] : -1;
    int l0 = lane, l1 = lane + 64;
    for (int i = 0; i < H; ++i) {
      const int64_t vi = i < 64 ? __shfl(v0, i, TG_WAVE) : __shfl(v1, i - 64, TG_WAVE);
      if (vi == v0 && i > l0) l0 = i;
      if (vi == v1 && i > l1) l1 = i;
    }
    const unsigned long long m0 = __ballot(l0 == lane);
    const unsigned long long m1 = __ballot(lane + 64 < H && l1 == lane + 64);
    auto right_of = [&](int last) {  // distinct ids whose last occurrence lies to the right of column `last`
      if (last < 63) return __popcll(m0 >> (last + 1)) + __popcll(m1);
      if (last == 63) return __popcll(m1);
      return last >= 127 ? 0 : __popcll(m1 >> (last - 64 + 1));
    };
    out[r * H + lane] = (v0 == 0) ? 0 : 1 + right_of(l0);
    if (lane + 64 < H) out[r * H + 64 + lane] = (v1 == 0) ? 0 : 1 + right_of(l1);
  }
}

}  // namespace tg

using namespace tg;

static int sample_args_ok(const tg_tcsr* g, int64_t Q, const void* a, const void* b, int K, const void* o1,
                          const void* o2, const void* o3) {
  if (!g || Q < 0 || K <= 0) return 0;
  if (Q > 0 && (!a || !b || !o1 || !o2 || !o3)) return 0;
  return 1;
}

extern "C" int tg_sample_recent_edges(const tg_tcsr* g, int64_t Q, const int64_t* nids, const double* ts, int32_t K,
                                      int64_t* o_nbr, int64_t* o_eid, float* o_ts, int64_t* o_dir, uint8_t* mark,
                                      void* stream) {
  if (!sample_args_ok(g, Q, nids, ts, K, o_nbr, o_eid, o_ts)) return TG_EINVAL;
  if (Q == 0) return TG_OK;
  if (K <= 16) {
    const unsigned grid = flat_grid(Q, 256 / 16);
    hipLaunchKernelGGL(k_sample_recent_edges<16>, dim3(grid), dim3(256), 0, as_stream(stream), *g, Q, nids, ts, K,
                       o_nbr, o_eid, o_ts, o_dir, mark);
  } else {
    const unsigned grid = flat_grid(Q, 256 / 64);
    hipLaunchKernelGGL(k_sample_recent_edges<64>, dim3(grid), dim3(256), 0, as_stream(stream), *g, Q, nids, ts, K,
                       o_nbr, o_eid, o_ts, o_dir, mark);
  }
  return check_launch("tg_sample_recent_edges");
}

namespace tg {
__global__ void k_mark_lists(int64_t Q, int K, const int64_t* __restrict__ nids, const int64_t* __restrict__ nbr,
                             uint8_t* __restrict__ mark) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Q * (K + 1); i += (int64_t)gridDim.x * blockDim.x)
    mark[i < Q ? nids[i] : nbr[i - Q]] = 1;  // byte flags, plain stores (padding id 0 included, as the sampler does)
}
}  // namespace tg
int tg::sample_nodes_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const double* ts, int32_t K, int64_t* o_nbr,
                            int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st) {
  if (Q <= 0) return TG_OK;
  if (K > TG_WAVE) return TG_EUNSUPPORTED;
  hipLaunchKernelGGL(k_sample_recent_nodes, dim3(flat_grid(Q, 4)), dim3(256), 0, st, *g, Q, nids, ts, K, o_nbr, o_eid, o_ts,
                     (int64_t*)nullptr);
  if (mark)
    hipLaunchKernelGGL(k_mark_lists, dim3(flat_grid(Q * (K + 1), 256)), dim3(256), 0, st, Q, (int)K, nids,
                       (const int64_t*)o_nbr, mark);
  return check_launch("sample_nodes");
}

// second hop of a two-layer step with --strategy recent_nodes (data_loader.py:128-131: the graph's own strategy at the
// neighbours' float32 timestamps)
int tg::sample_nodes_f32_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const float* ts, int32_t K, int64_t* o_nbr,
                                int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st) {
  if (Q <= 0) return TG_OK;
  if (K > TG_WAVE) return TG_EUNSUPPORTED;
  hipLaunchKernelGGL(k_sample_recent_nodes, dim3(flat_grid(Q, 4)), dim3(256), 0, st, *g, Q, nids, (const double*)nullptr, K,
                     o_nbr, o_eid, o_ts, (int64_t*)nullptr, ts);
  if (mark)
    hipLaunchKernelGGL(k_mark_lists, dim3(flat_grid(Q * (K + 1), 256)), dim3(256), 0, st, Q, (int)K, nids,
                       (const int64_t*)o_nbr, mark);
  return check_launch("sample_nodes_f32");
}

int tg::sample_edges_f32_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const float* ts, int32_t K, int64_t* o_nbr,
                                int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st) {
  if (Q <= 0) return TG_OK;
  if (K <= 16)
    hipLaunchKernelGGL(k_sample_recent_edges<16>, dim3(flat_grid(Q, 256 / 16)), dim3(256), 0, st, *g, Q, nids,
                       (const double*)nullptr, K, o_nbr, o_eid, o_ts, (int64_t*)nullptr, mark, ts);
  else
    hipLaunchKernelGGL(k_sample_recent_edges<64>, dim3(flat_grid(Q, 256 / 64)), dim3(256), 0, st, *g, Q, nids,
                       (const double*)nullptr, K, o_nbr, o_eid, o_ts, (int64_t*)nullptr, mark, ts);
  return check_launch("sample_edges_f32");
}

extern "C" int tg_sample_recent_nodes(const tg_tcsr* g, int64_t Q, const int64_t* nids, const double* ts, int32_t K,
                                      int64_t* o_nbr, int64_t* o_eid, float* o_ts, int64_t* o_dir, void* stream) {
  if (!sample_args_ok(g, Q, nids, ts, K, o_nbr, o_eid, o_ts)) return TG_EINVAL;
  if (K > TG_WAVE) return TG_EUNSUPPORTED;
  if (Q == 0) return TG_OK;
  hipLaunchKernelGGL(k_sample_recent_nodes, dim3(flat_grid(Q, 4)), dim3(256), 0, as_stream(stream), *g, Q, nids, ts,
                     K, o_nbr, o_eid, o_ts, o_dir);
  return check_launch("tg_sample_recent_nodes");
}

namespace tg {
__global__ void k_mark_queries(int64_t Q, int K, const int64_t* __restrict__ nids, const int64_t* __restrict__ nbr,
                               uint8_t* __restrict__ mark) {
  const int64_t n = Q * (K + 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i < Q ? nids[i] : nbr[i - Q];
    mark[v] = 1;
  }
}
// uniform sampler over prepared query arrays (the fused step, tg_step_io.strategy = 2); marks queries and neighbours in
// `mark` when given.  ONE wavefront walks the queries: the reference consumes its random stream in query order
int sample_uniform_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const double* ts, int32_t K, uint32_t* mt_state,
                          int64_t* o_nbr, int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st) {
  if (!mt_state) return TG_EINVAL;
  if (K > 16) return TG_EUNSUPPORTED;  // (numpy's argsort is a stable insertion sort only up to 16 elements)
  hipLaunchKernelGGL(k_sample_uniform, dim3(1), dim3(64), 0, st, *g, Q, nids, ts, (int)K, mt_state, o_nbr, o_eid, o_ts,
                     (int64_t*)nullptr);
  if (mark) hipLaunchKernelGGL(k_mark_queries, dim3(flat_grid(Q * (K + 1), 256)), dim3(256), 0, st, Q, (int)K, nids, o_nbr, mark);
  return check_launch("sample_uniform");
}
// second hop of a two-layer step with --strategy uniform (data_loader.py:128-131: the graph's own strategy at the neighbours'
// float32 timestamps): the stream goes on where the first hop left it, per non-empty query in slot order
int sample_uniform_f32_launch(const tg_tcsr* g, int64_t Q, const int64_t* nids, const float* ts, int32_t K, uint32_t* mt_state,
                              int64_t* o_nbr, int64_t* o_eid, float* o_ts, uint8_t* mark, hipStream_t st) {
  if (!mt_state) return TG_EINVAL;
  if (K > 16) return TG_EUNSUPPORTED;
  if (Q <= 0) return TG_OK;
  hipLaunchKernelGGL(k_sample_uniform, dim3(1), dim3(64), 0, st, *g, Q, nids, (const double*)nullptr, (int)K, mt_state, o_nbr,
                     o_eid, o_ts, (int64_t*)nullptr, ts);
  if (mark) hipLaunchKernelGGL(k_mark_queries, dim3(flat_grid(Q * (K + 1), 256)), dim3(256), 0, st, Q, (int)K, nids, o_nbr, mark);
  return check_launch("sample_uniform_f32");
}
}  // namespace tg

extern "C" int tg_sample_uniform(const tg_tcsr* g, int64_t Q, const int64_t* nids, const double* ts, int32_t K,
                                 uint32_t* mt_state, int64_t* o_nbr, int64_t* o_eid, float* o_ts, int64_t* o_dir,
                                 void* stream) {
  if (!sample_args_ok(g, Q, nids, ts, K, o_nbr, o_eid, o_ts) || !mt_state) return TG_EINVAL;
  // numpy sorts the K draws with argsort: insertion sort (stable) only up to 16
  // elements; beyond that the tie order is implementation defined.
  if (K > 16) return TG_EUNSUPPORTED;
  if (Q == 0) return TG_OK;
  hipLaunchKernelGGL(k_sample_uniform, dim3(1), dim3(64), 0, as_stream(stream), *g, Q, nids, ts, K, mt_state, o_nbr,
                     o_eid, o_ts, o_dir);
  return check_launch("tg_sample_uniform");
}

extern "C" int tg_rand_edge_pairs(uint32_t* mt_state, int64_t n_src, int64_t n_dst, int64_t count, const int64_t* src_list,
                                  const int64_t* dst_list, int64_t* out_src, int64_t* out_dst, void* stream) {
  if (!mt_state || n_src <= 0 || n_dst <= 0 || count < 0) return TG_EINVAL;
  if (n_src > 0xFFFFFFFFll || n_dst > 0xFFFFFFFFll) return TG_EUNSUPPORTED;  // 64-bit draws: no node table is that long
  if (count == 0) return TG_OK;
  if (!out_src || !out_dst) return TG_EINVAL;
  hipLaunchKernelGGL(k_rand_edge_pairs, dim3(1), dim3(64), 0, as_stream(stream), mt_state, (uint32_t)(n_src - 1),
                     (uint32_t)(n_dst - 1), count, src_list, dst_list, out_src, out_dst);
  return check_launch("tg_rand_edge_pairs");
}

extern "C" int tg_hits(int64_t B, int32_t K, const int64_t* center, const int64_t* nbr, float* out, void* stream) {
  if (B < 0 || K <= 0) return TG_EINVAL;
  if (B == 0) return TG_OK;
  if (!center || !nbr || !out) return TG_EINVAL;
  hipLaunchKernelGGL(k_hits, dim3(flat_grid(B * K, 256)), dim3(256), 0, as_stream(stream), B * K, K, center, nbr, out);
  return check_launch("tg_hits");
}

extern "C" int tg_anonymized_reindex(int64_t n, int32_t H, const int64_t* in, int64_t* out, void* stream) {
  if (n < 0 || H <= 0) return TG_EINVAL;
  if (H > 2 * TG_WAVE) return TG_EUNSUPPORTED;
  if (n == 0) return TG_OK;
  if (!in || !out) return TG_EINVAL;
  if (H <= TG_WAVE)
    hipLaunchKernelGGL(k_anon_reindex, dim3(flat_grid(n, 4)), dim3(256), 0, as_stream(stream), n, H, in, out);
  else
    hipLaunchKernelGGL(k_anon_reindex2, dim3(flat_grid(n, 4)), dim3(256), 0, as_stream(stream), n, H, in, out);
  return check_launch("tg_anonymized_reindex");
}

// ---------------------------------------------------------------------------------
// Negative sampling stream of RandEdgeSampler (data_loader.py:283-313) for a whole batch.
// The reference draws, per event, `rng.randint(0, n_src, 1)` then `rng.randint(0, n_dst, 1)` on a
// numpy legacy RandomState: MT19937 32-bit outputs under the smallest covering bit mask, redrawn
// while above the range (rk_random_uint64, range < 2^32).  Same state in, same state out as `count`
// calls of RandEdgeSampler.sample(1).  Host routine: the stream is sequential by construction.
// ---------------------------------------------------------------------------------
namespace tg {
struct HostMt {
  uint32_t* key;
  uint32_t pos;
  uint32_t next() {
    constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
    if (pos >= 624) {
      int kk = 0;
      for (; kk < 227; ++kk) {
        const uint32_t y = (key[kk] & UP) | (key[kk + 1] & LO);
        key[kk] = key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
      }
      for (; kk < 623; ++kk) {
        const uint32_t y = (key[kk] & UP) | (key[kk + 1] & LO);
        key[kk] = key[kk - 227] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
      }
      const uint32_t y = (key[623] & UP) | (key[0] & LO);
      key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
      pos = 0;
    }
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  int64_t below(int64_t n) {  // randint(0, n): n >= 1
    const uint32_t rng = (uint32_t)(n - 1);
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = next() & mask) > rng) {
    }
    return (int64_t)v;
  }
};
}  // namespace tg

extern "C" int tg_rand_edge_pairs_host(uint32_t* mt_state, int64_t n_src, int64_t n_dst, int64_t count,
                                       int64_t* src_idx, int64_t* dst_idx) {
  if (!mt_state || n_src <= 0 || n_dst <= 0 || n_src > 0xffffffffLL || n_dst > 0xffffffffLL || count < 0) return TG_EINVAL;
  if (count && (!src_idx || !dst_idx)) return TG_EINVAL;
  tg::HostMt mt{mt_state, mt_state[624]};
  for (int64_t i = 0; i < count; ++i) {
    src_idx[i] = mt.below(n_src);
    dst_idx[i] = mt.below(n_dst);
  }
  mt_state[624] = mt.pos;
  return TG_OK;
}
