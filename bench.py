#!/usr/bin/env python3
"""Headline benchmark: processed interaction-events/sec through the TIGER event-batch hot
path (temporal sampling -> mailbox consume + GRU -> temporal attention -> memory/mailbox
write-back), BASELINE.json configs[1]: JODIE-Wikipedia-shaped stream, d=172, batch=1024,
msg_src=left upd_src=left, on 1 (or N) MI355X.

  python bench.py --gpus N --steps K --warmup W

A step = one batch of B events through tg_stream_step.  The whole synthetic stream is
resident in HBM before the timed region.  Order of a run: PREROLL untimed batches bring the
memories / mailbox to steady state (independent of --warmup: the number of involved nodes per
batch keeps growing for ~100 batches), W untimed warm-up steps, then the timed region replays
a captured hipGraph of one step K times (the step reads its batch at a device-side offset and
advances it).  Per-stage HIP-event times are then taken on the NEXT unseen batches of the
stream (the state keeps moving forward: no batch is replayed).
Rank 0 prints ONE JSON line (see the task contract) including
  roofline               - the dominant kernel of the step, timed live with HIP events, priced with the
                           algorithmic flops / bytes of SURVEY.md s8(d) and the measured U / O / P of the run;
  roofline_memory_gather - the memory-gather kernel OF THE TIMED STEP (eager direct form: the attention core, which gathers the
                           involved rows itself) against the HBM roofline on its COMPULSORY bytes, the design's and the
                           PMC bytes beside it, plus SURVEY's full bytes_gather over gather + updater;
  roofline_sampler       - the sampler launch on the HBM roofline (SURVEY s8 d bytes_samp) where it has a launch of its own;
  c5s_leg                - (default C2 run only) a short run of the HBM-roofline configuration (10 M nodes,
                           d=256, B=65536): at C2 every table is cache resident, so the HBM claim is made there;
  cpu_baseline           - the CPU oracle (oracle/tiger_oracle.py, "port") on a bounded sample of the same
                           workload on this host's cores (thread count chosen by a quick sweep).
With --gpus N > 1 and no torchrun environment the script starts its own N ranks.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured)

# BASELINE.json configs[1]; Wikipedia: 8227 users, 1000 items, 157474 events
C2 = dict(name='C2 JODIE-Wikipedia-shaped synthetic', n_u=8227, n_i=1000, E=157474, T=2.68e6, d=172, K=10, B=1024,
          msg_src='left', upd_src='left')
# the other BASELINE configs: parity-test shapes, selectable for profiling runs only
WORKLOADS = {
    'c1': dict(C2, name='C1 JODIE-Wikipedia-shaped synthetic, the reference default batch', B=200, upd_src='right'),
    'c2': C2,
    'c3': dict(name='C3 JODIE-Reddit-shaped synthetic, static restarter, restart_prob=0.01 (lazy restarts in the step)',
               n_u=10000, n_i=984, E=672447, T=2.68e6, d=172, K=10, B=4096, msg_src='left', upd_src='right',
               restart_prob=0.01),
    'c4': dict(name='C4 JODIE-LastFM-shaped synthetic (no feature tables)', n_u=980, n_i=1000, E=1293103, T=1.37e8, d=100,
               K=10, B=8192, msg_src='left', upd_src='right', no_feats=True),
    # C5 scaled to one GPU-box host: 10 M nodes as in BASELINE, 4 M events (the state tables are full size:
    # 2 x 10.2 GB memories + 41 GB mailbox), d=256, B=65536 - the HBM-roofline configuration
    'c5s': dict(name='C5 synthetic 10M nodes d=256 B=65536 (stream shortened to 4M events)', n_u=9000000, n_i=1000000,
                E=4000000, T=4.0e6, d=256, K=10, B=65536, msg_src='left', upd_src='right', no_feats=True,
                integer_ts=False),
    # BASELINE configs[4] as written (SURVEY.md s8 d): 10^8 events over 10 M nodes, the stream from the counter-based
    # generator - a run generates the prefix it processes (every rank the same slice, by construction), at the event
    # rate of the full stream (T / E = 1); the first 10 % of the stream is warm-up (19 global batches at 8 x 65536)
    'c5': dict(name='C5 synthetic 10^8 events / 10M nodes d=256 B=65536 (counter-based stream; the processed prefix is generated)',
               n_u=9000000, n_i=1000000, E=100000000, T=1.0e8, d=256, K=10, B=65536, msg_src='left', upd_src='right',
               no_feats=True, integer_ts=False, counter_stream=True),
}


def _mix64(x):
    """splitmix64 finaliser on uint64 arrays: the counter-based generator's hash"""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def counter_uniform(seed, lane, lo, hi):
    """uniform [0, 1) doubles for stream positions [lo, hi): a pure function of (seed, lane, position)"""
    with np.errstate(over='ignore'):
        idx = np.arange(lo, hi, dtype=np.uint64)
        key = np.uint64((seed * 0x9E3779B97F4A7C15 + lane * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
        bits = _mix64(_mix64(idx + key) ^ np.uint64(0xA0761D6478BD642F))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_stream_counter(n_u, n_i, E_total, T_total, lo, hi, seed=0):
    """Events [lo, hi) of the SURVEY.md s8(d) C5 stream from a counter-based generator: every quantity is a pure
    function of (seed, position), so any rank (or shard) generates exactly the slice it needs - identical wherever it
    is generated, no RandomState walked over the whole 10^8-event stream.  Same marginals as make_stream: Zipf(0.8)
    users / Zipf(1.0) items by inverse CDF, timestamps increasing with the position (one uniform draw inside each of
    the E_total equal slots of [0, T_total)), uniform negatives."""
    cu = np.cumsum(1.0 / np.arange(1, n_u + 1) ** 0.8)
    ci = np.cumsum(1.0 / np.arange(1, n_i + 1) ** 1.0)
    src = np.minimum(np.searchsorted(cu, counter_uniform(seed, 1, lo, hi) * cu[-1], side='right'), n_u - 1).astype(np.int64) + 1
    dst = np.minimum(np.searchsorted(ci, counter_uniform(seed, 2, lo, hi) * ci[-1], side='right'), n_i - 1).astype(np.int64) + 1 + n_u
    ts = (np.arange(lo, hi, dtype=np.float64) + counter_uniform(seed, 3, lo, hi)) * (T_total / E_total)
    neg = (counter_uniform(seed, 4, lo, hi) * n_i).astype(np.int64) + 1 + n_u
    return dict(src=src, dst=dst, ts=ts, eids=np.arange(lo + 1, hi + 1, dtype=np.int64), neg=neg,
                n_nodes=n_u + n_i + 1, efeats=None)


def make_stream(n_u, n_i, E, T, seed=0, d_e=172, integer_ts=True, with_efeats=True, counter=False):
    """SURVEY.md s8(d) generator: bipartite ids (0 = padding, users 1..n_u, items after),
    Zipf(0.8) users, Zipf(1.0) items, sorted uniform timestamps (floored: duplicates occur),
    eid = 1..E, N(0,1) edge features with row 0 = 0, one pre-drawn negative per event.
    counter: the first E events of the counter-based stream (make_stream_counter; no edge features)."""
    if counter:
        assert not with_efeats and not integer_ts
        return make_stream_counter(n_u, n_i, E, T, 0, E, seed)
    rs = np.random.RandomState(seed)
    pu = 1.0 / np.arange(1, n_u + 1) ** 0.8
    pi = 1.0 / np.arange(1, n_i + 1) ** 1.0
    src = rs.choice(n_u, E, p=pu / pu.sum()).astype(np.int64) + 1
    dst = rs.choice(n_i, E, p=pi / pi.sum()).astype(np.int64) + 1 + n_u
    ts = np.sort(rs.uniform(0, T, E))
    if integer_ts:
        ts = np.floor(ts)
    neg = rs.randint(n_u + 1, n_u + n_i + 1, E).astype(np.int64)
    out = dict(src=src, dst=dst, ts=ts.astype(np.float64), eids=np.arange(1, E + 1, dtype=np.int64), neg=neg,
               n_nodes=n_u + n_i + 1, efeats=None)
    if with_efeats:
        ef = rs.standard_normal((E + 1, d_e)).astype(np.float32)
        ef[0] = 0
        out['efeats'] = ef
    return out


def build_models(stream, d, K, msg_src, upd_src, restarter='static', hist_len=40, with_oracle=False, device='cuda:0',
                 zero_nfeats=True, seed=0, dropout=0.1):
    """HIP model (reference initialisers under torch.manual_seed) and, optionally, the CPU
    oracle carrying the very same weights."""
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.model.feature_getter import NumericalFeature
    from www2023tiger_amd.model.restarters import SeqRestarter, StaticRestarter
    from www2023tiger_amd.model.tiger import TIGER
    dev = torch.device(device)
    n_nodes = stream['n_nodes']
    g = Graph.from_arrays(stream['src'], stream['dst'], stream['ts'], stream['eids'], strategy='recent_edges', seed=0,
                          max_node_id=n_nodes - 1, device=dev)
    nfeats = np.zeros((n_nodes, d), dtype=np.float32) if zero_nfeats else None  # JODIE node features are all zero
    efeats = stream['efeats']
    torch.manual_seed(seed)
    with torch.device(dev):  # parameters and the (possibly tens of GB of) state tables are born on the GPU
        fg = NumericalFeature(None if nfeats is None else torch.from_numpy(nfeats).to(dev),
                              None if efeats is None else torch.from_numpy(efeats).to(dev), dim=d, device=dev)
        fg.n_nodes, fg.n_edges = n_nodes, len(stream['src'])
        if restarter == 'seq':
            rst = SeqRestarter(raw_feat_getter=fg, graph=g, hist_len=hist_len, n_head=2, dropout=dropout)
        else:
            rst = StaticRestarter(raw_feat_getter=fg, graph=g)
        model = TIGER(raw_feat_getter=fg, graph=g, restarter=rst, n_neighbors=K, hit_type='bin', n_layers=1, n_head=2,
                      dropout=dropout, msg_src=msg_src, upd_src=upd_src)
        with torch.no_grad():  # non-trivial time-encoder phase so the cos path is exercised
            model.time_encoder.phase.uniform_(-0.5, 0.5)
    model = model.to(dev).eval()
    oracle = None
    if with_oracle:
        from oracle import tiger_oracle as O
        og = O.OracleGraph(stream['src'], stream['dst'], stream['ts'], stream['eids'], max_node_id=n_nodes - 1)
        params = {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}
        oracle = O.OracleTIGER(params, og, n_nodes=n_nodes, dim=d, nfeats=nfeats, efeats=efeats, n_neighbors=K,
                               msg_src=msg_src, upd_src=upd_src, restarter=restarter, hist_len=hist_len)
    return model, oracle



MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32 MFMA (MI355X_MICROARCH.md)
PREROLL = 150                 # untimed batches before --warmup (state reaches steady U / O / P after ~100 at C2)
C5S_PREROLL = 600             # ... of the C5-shaped leg (39 M events).  The involved set per batch never stops growing there - a
                              # Zipf(0.8) tail over 9 M users keeps delivering nodes whose histories are shorter than K - but
                              # its growth over the timed region is below 1 % by then; the sizes at the middle / end of the
                              # pre-roll and after the timed region are in the leg's config (involved_start_end)
REFERENCE_MEASURED = ('5730 events/s @ 8 cores: the reference\'s own CPU stream (collate + contrast_learning, no_grad), '
                      'd=172 B=1024, measured by importing it in the survey container (BASELINE.md s2)')


def cpu_baseline(stream, cfg, model, budget_s=45.0, batches=100, warm=10):
    """Oracle (CPU restatement, parity-pinned to the reference) on the first batches of the same stream:
    collate + STEP 1-6 per batch.  The intra-op thread count is chosen by a quick sweep (the box may expose
    more hardware threads than its share of cores: more threads than that is slower)."""
    from oracle import tiger_oracle as O
    og = O.OracleGraph(stream['src'], stream['dst'], stream['ts'], stream['eids'], max_node_id=stream['n_nodes'] - 1)
    params = {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}
    nfeats = np.zeros((stream['n_nodes'], cfg['d']), dtype=np.float32)
    orc = O.OracleTIGER(params, og, n_nodes=stream['n_nodes'], dim=cfg['d'], nfeats=nfeats, efeats=stream['efeats'],
                        n_neighbors=cfg['K'], msg_src=cfg['msg_src'], upd_src=cfg['upd_src'], restarter='static')
    B = cfg['B']
    pos = [0]

    def run(n):
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                sl = slice(pos[0] * B, (pos[0] + 1) * B)
                pos[0] += 1
                a = [stream[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
                orc.stream_step(*a, O.collate(og, a[0], a[1], a[2], a[3], cfg['K'], 'static'))
        return time.perf_counter() - t0

    run(2)
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu})
    sweep = {}
    for t in cands:  # 2 batches per candidate
        torch.set_num_threads(t)
        sweep[t] = 2 * B / run(2)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    run(max(0, warm - 2 - 2 * len(cands)))
    done, elapsed = 0, 0.0
    while done < batches and elapsed < budget_s and (pos[0] + 1) * B <= len(stream['src']):
        elapsed += run(1)
        done += 1
    return dict(value=done * B / elapsed, unit='events/s', cores=best, kind='port',
                sample=f'oracle/tiger_oracle.py, {done} timed batches of B={B} after {pos[0] - done} warm-up batches '
                       f'(collate + STEP 1-6), {elapsed:.1f} s; thread sweep (events/s on 2 batches): '
                       + ', '.join(f'{t}: {v:.0f}' for t, v in sweep.items()),
                host_threads_visible=ncpu, reference_measured=REFERENCE_MEASURED)


def profile_stages(model, buf, steps, record=True):
    """Eager steps on the next batches of the stream with the library's per-stage HIP-event timer attached
    (events are recorded on the stream the kernels are launched on)."""
    from www2023tiger_amd._lib import lib
    n = lib.tg_profiler_num_stages()
    names = [lib.tg_profiler_stage_name(i).decode() for i in range(n)]
    prof = lib.tg_profiler_create()
    assert prof, 'tg_profiler_create failed'
    buf.attach_profiler(prof)
    acc = np.zeros(n)
    counts = np.zeros(4)
    ms = (C.c_float * n)()
    # kernel-bound durations (event pairs bound to the dispatches of the step's main kernels: what rocprofv3 reports)
    nk = lib.tg_profiler_num_kernel_slots()
    kms, knames = (C.c_float * nk)(), (C.c_char_p * nk)()
    kacc, khits = np.zeros(nk), np.zeros(nk)
    for _ in range(steps):
        model.launch_step(buf)
        rc = lib.tg_profiler_read(prof, ms)
        assert rc == 0
        acc += np.array(ms[:])
        assert lib.tg_profiler_kernel_ms(prof, kms, knames) == 0
        for i in range(nk):
            if kms[i] >= 0:
                kacc[i] += kms[i]
                khits[i] += 1
        counts += buf.counts.cpu().numpy()
    if record:  # (side passes - set sizes, the copy form's gather - do not replace the main pass's kernels)
        KERNEL_MS.clear()
        for i in range(nk):
            if khits[i] and knames[i]:
                KERNEL_MS[lib.tg_profiler_kernel_slot_name(i).decode()] = (kacc[i] / khits[i], norm_kernel(knames[i].decode()))
    buf.attach_profiler(None)
    lib.tg_profiler_destroy(prof)
    return names, acc / steps, counts / steps


KERNEL_MS = {}   # slot -> (kernel-bound average ms, launch expression) of the last profile_stages pass
SLOT_OF_STAGE = {'attn_core(gather+softmax)': 'attn_core', 'attn_gemm_fc1': 'fc1', 'attn_gemm_fc2': 'fc2',
                 'eager_updater(gru)': 'updater', 'apply_messages(gru)': 'updater', 'eager_query_rows(G)': 'query_rows',
                 'sample_recent_edges': 'collate(sampler+centres)', 'writeback_phase1': 'writeback',
                 'gather_right_memory': 'gather'}


def norm_kernel(expr):
    """'(k_gemm_ks16<WbRider, 3, 3, 4>)' / 'tg::k_gemm_ks16<tg::WbRider, 3, 3, 4, tg::NoSecond>' -> comparable form"""
    e = expr.strip().strip('()').replace('tg::', '').replace(' ', '')
    return e


def stage_work(cfg, U, O_, P, eager, fused, n_nodes, E, tile=False, gtab=False):
    """Algorithmic work per launch of every stage (SURVEY.md s8 d, with the measured U = involved, O = with a
    pending message, P = unique positive nodes of the run): name -> (flops or None, HBM bytes, kernel name)."""
    B, K, d = cfg['B'], cfg['K'], cfg['d']
    Q, d_e = 3 * B, d
    fe = 0 if cfg.get('no_feats') else 1
    mw = 3 * d + d_e               # mailbox row width
    # without an edge table the edge segment of mailbox rows and key rows is zeros: the pre-multiplied attention weights
    # drop its columns and the updater skips the k-tiles that lie inside it (no flops are counted for skipped zeros)
    kvw, nh = 2 * d + (d_e if (fe or not fused) else 0), 2
    nk = nh * kvw
    deg = max(2.0, 2.0 * E / n_nodes)
    upd_rows = P if eager else O_  # rows the updater runs on
    mw_mul = mw if fe else mw - 32 * max(0, (2 * d + d_e) // 32 - (2 * d + 31) // 32)
    gru = (2.0 * upd_rows * 3 * d * (mw_mul + d), upd_rows * (4 * mw_mul + 4) + upd_rows * (4 * d + 4) + upd_rows * 4 * d)
    w = {
        'sample_recent_edges': (None, Q * (8 * np.ceil(np.log2(deg + 1)) + K * 28) + 5 * 8 * B, 'tg::k_sample_batch<16>'),
        'unique_compact': (None, n_nodes + 8 * U + 12 * O_, 'tg::k_bm_small' if n_nodes <= 65536 else 'tg::k_bm_emit'),
        # eager: every involved row is a live copy (pending-or-right row -> reprs); lazy: only rows without a pending message
        'gather_right_memory': (None, 2.0 * 4 * d * (U if eager else max(U - O_, 0)) + 12 * O_, 'tg::k_consume_gather_check'),
        'apply_messages(gru)': gru + ('tg::k_gru',),
        'eager_updater(gru)': gru + ('tg::k_gru_direct16 / tg::k_gru_direct / tg::k_gru',),
        'attn_centres+qconst': (None, Q * 4 * d * (2 + fe), 'tg::k_attn_centres'),
        'attn_core(gather+softmax)': (None, U * 4 * d * (1 + fe) + Q * K * 4 * d_e * fe + 2.0 * Q * nk * 4 + Q * K * 20,
                                      'tg::k_attn_core'),
        'attn_gemm_fc2': (2.0 * Q * d * d, Q * d * 8 + d * d * 4, 'tg::k_gemm_direct_r / tg::k_gemm_direct / tg::k_gemm_r / tg::k_gemm'),
        'writeback_phase0': (None, P * (4 * mw + 4) + P * 4 * d * 2 + 2 * B * 4 * d + B * 4 * d_e * fe, 'tg::k_writeback<0>'),
        'writeback_phase1': (None, 2 * P * (4 * d + 5) + n_nodes, 'tg::k_writeback<1>'),
    }
    if fused and tile:
        # the whole attention block is ONE launch (k_attn_tile): G and S stay in LDS.  flops: the three products;
        # compulsory bytes: the unique neighbour rows (+ their node features), the edge-feature rows, the neighbour
        # lists, centre rows in, embeddings out, every weight once
        wts = (nk * d + nk + d * (nk + d) + 2 * d + d * d + d) * 4
        w['attn_core(gather+softmax)'] = (2.0 * Q * (nk * d + d * (nk + d) + d * d),
                                          U * 4 * d * (1 + fe) + Q * K * 4 * d_e * fe + Q * K * 20 + 2.0 * Q * d * 4 + wts,
                                          'tg::k_attn_tile')
    elif fused:
        w['attn_gemm_q'] = (2.0 * Q * nk * d, Q * d * 4 + Q * nk * 4 + nk * d * 4, 'tg::k_gemm')           # G = c Wqk^T + gconst
        if gtab:  # eager query rows: the product runs on the P positive nodes at the end of the step instead of on Q centres
            w['eager_query_rows(G)'] = (2.0 * P * nk * d, P * d * 4 * (2 + fe) + P * nk * 4 + nk * d * 4,
                                        'tg::k_gemm_direct_r / tg::k_gemm_direct / tg::k_gemm_astat_r / tg::k_gemm_astat / tg::k_gemm_rb / tg::k_gemm')
        w['attn_gemm_fc1'] = (2.0 * Q * d * (nk + d), Q * (nk + d) * 4 + Q * d * 4 + d * (nk + d) * 4, 'tg::k_gemm_ks16 / tg::k_gemm_sk / tg::k_gemm_rb / tg::k_gemm')
    else:
        w['attn_gemm_q'] = (2.0 * Q * 2 * d * d, Q * d * 4 + Q * 2 * d * 4, 'tg::k_gemm')
        w['attn_gemm_g'] = (2.0 * Q * kvw * 2 * d, Q * 2 * d * 4 + Q * nk * 4, 'tg::k_gemm')
        w['attn_gemm_v'] = (2.0 * Q * kvw * 2 * d, Q * nk * 4 + Q * 2 * d * 4, 'tg::k_gemm')
        w['attn_gemm_out'] = (2.0 * Q * 2 * d * 2 * d, Q * 2 * d * 8, 'tg::k_gemm')
        w['attn_gemm_fc1'] = (2.0 * Q * d * 3 * d, Q * 3 * d * 4 + Q * d * 4, 'tg::k_gemm')
    return w


TRAFFIC_SOURCE = {}


def load_traffic(tag):
    """HBM / fabric bytes per launch from the committed PMC passes (profiles/r04_hbm_traffic_<tag>.json, else an earlier
    round's: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same command, tools/pmc_traffic.py; with the rocprofv3
    --kernel-trace average duration of the same kernels beside them when recorded); {} if not recorded.
    A stored, builder-run measurement - the profiler cannot wrap the driver's run - and named as such in the line."""
    for rnd in ('r05', 'r04', 'r03', 'r02'):
        path = os.path.join('profiles', f'{rnd}_hbm_traffic_{tag}.json')
        try:
            k = json.load(open(os.path.join(ROOT, path)))['kernels']
            TRAFFIC_SOURCE[tag] = path + ' (builder-run PMC passes of the same command, not measured in this run)'
            return k
        except (OSError, ValueError, KeyError):
            continue
    return {}


def kernel_entry(traffic, kname, launched=None):
    """The stored PMC / rocprof entry of a stage's kernel.  launched: the launch expression the profiler recorded for the
    stage in THIS run (norm_kernel form) - the entry must be that very instantiation (a stored name may carry further,
    defaulted template arguments); without it (no kernel-bound timing) kname lists candidates separated by ' / ' in
    order of preference and the first base-name match counts."""
    if launched:
        base, args = (launched.split('<', 1) + [''])[:2]
        args = args.rstrip('>')
        for k, v in traffic.items():
            kb, ka = (norm_kernel(k).split('<', 1) + [''])[:2]
            if kb == base and (ka.rstrip('>') == args or ka.startswith(args + ',')):
                return k, v
        return None, None
    for cand in kname.split(' / '):
        base = cand.strip().split('<')[0].split(' ')[0]
        for k, v in traffic.items():
            if k.split('<')[0] == base:
                return k, v
    return None, None


def kernel_traffic(traffic, kname, launched=None):
    return (kernel_entry(traffic, kname, launched)[1] or {}).get('bytes_per_launch')


def roofline_of(name, ms, work, traffic, overhead=0.0):
    """ms: HIP-event interval around the stage's launch; overhead: what an EMPTY interval of the same pass measures (the
    event pair itself, ~5 us on this stack) - the launch duration is the difference, which is what the rocprofv3 average
    of the kernel agrees with (profiles/); both are reported"""
    flops, nbytes, kname = work.get(name, (None, None, name))
    raw_ms = float(ms)
    # the launch's duration: the event pair BOUND to the dispatch (its begin / end timestamps, as rocprofv3 reports them)
    # when the profiler timed this stage's kernel; else the interval less an empty pair, never less than 0.8 of it (the
    # subtraction over-corrects: part of the records overlaps the kernel's start)
    kb = KERNEL_MS.get(SLOT_OF_STAGE.get(name, ''))
    launched = kb[1] if kb else None
    ms = float(kb[0]) if kb else max(raw_ms - overhead, 0.8 * raw_ms)
    timing = 'kernel-bound HIP events (hipExtLaunchKernelGGL start / stop)' if kb else 'HIP-event interval less an empty pair'
    t_s = ms * 1e-3
    ekey, entry = kernel_entry(traffic, kname, launched)
    tr = (entry or {}).get('bytes_per_launch')
    extra = dict(avg_ms=float(ms), timing=timing, avg_ms_event_interval=raw_ms, event_pair_ms=float(overhead),
                 launched=launched, traffic=tr, traffic_kernel=ekey,
                 rocprof_avg_ms_stored=(entry or {}).get('rocprof_avg_us', None) and entry['rocprof_avg_us'] * 1e-3)
    if tr and nbytes:
        extra['traffic_over_algorithmic'] = tr / nbytes
    if flops:
        ach = flops / t_s / 1e12
        return dict(bound='mfma', kernel=name, device_kernel=launched or kname, achieved=ach, peak=MFMA_F32_PEAK_TFLOPS,
                    unit='TFLOP/s', frac=ach / MFMA_F32_PEAK_TFLOPS, algorithmic_flops=float(flops),
                    algorithmic_bytes=float(nbytes), hbm_gbs=nbytes / t_s / 1e9,
                    hbm_frac=nbytes / t_s / 1e9 / HBM_PEAK_GBS, **extra)
    if nbytes:
        ach = nbytes / t_s / 1e9
        return dict(bound='hbm', kernel=name, device_kernel=launched or kname, achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=ach / HBM_PEAK_GBS, algorithmic_bytes=float(nbytes), **extra)
    return dict(bound='hbm', kernel=name, achieved=None, peak=HBM_PEAK_GBS, unit='GB/s', frac=None, **extra)


def run_stream_leg(cfg, args, preroll, warmup, steps, n_prof, traffic_tag, want_cpu=False):
    """One single-GPU measurement of the streaming step on workload `cfg`; returns the pieces of the JSON line."""
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    B, K, d = cfg['B'], cfg['K'], cfg['d']
    # the timed region is K steps, once (the contract); `reps` further, separately timed repetitions of the same K steps
    # on the following batches give its spread (a K = 20 region is ONE 1.7 ms graph replay)
    reps = args.repeats if (B <= 8192 and not args.no_graph) else 0
    n_batches = preroll + warmup + steps * (1 + reps) + n_prof + 8  # + 4 batches for the copy-form gather timing, + 2 for the set sizes of a lean run, + 2 spare
    E = max(cfg['E'], n_batches * B) if not cfg.get('counter_stream') else n_batches * B
    no_feats = bool(cfg.get('no_feats'))
    stream = make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=d,
                         integer_ts=cfg.get('integer_ts', True), with_efeats=not no_feats,
                         counter=bool(cfg.get('counter_stream')))
    model, _ = build_models(stream, d, K, cfg['msg_src'], cfg['upd_src'], restarter='static', device='cuda:0',
                            zero_nfeats=not no_feats)
    resident = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
    fused, eager = not args.no_fuse, not args.no_eager
    if fused:   # streaming inference, parameters fixed: pre-multiplied attention weights (tg_attn_fuse)
        model.fuse_attention()
    if eager:   # ... and the updater run once per stored message (TIGE.eager_updates)
        model.eager_updates()
    # prefetch: the next batch's sampler + centres ride on the step's last launch (tg_step_io.prefetch_state; honoured by
    # lean eager steps with eager query rows only); the self-check's second model runs without it
    buf = model.StepBuffers(model, B, False, resident=resident, prefetch=not args.no_prefetch)
    if args.eager_copy:
        buf.io.eager_copy = 1
    restart_prob = float(cfg.get('restart_prob', 0.0))
    n_trig = 0
    if restart_prob > 0:  # train_self_supervised.py:153: one uniform draw per batch, never before batch 0
        trig = (np.random.RandomState(1).rand(n_batches) < restart_prob).astype(np.uint8)
        trig[0] = 0
        with torch.no_grad():  # trained surrogate rows (the reference initialises the tables with zeros)
            torch.manual_seed(1)
            model.restarter_fn.left_emb.weight.normal_(0.0, 0.5)
            model.restarter_fn.right_emb.weight.normal_(0.0, 0.5)
        buf.enable_lazy_restart(model, trig)
        n_trig = int(trig[preroll + warmup:preroll + warmup + steps].sum())
    _ = model.graph.tcsr, model.model_struct()  # lazy device-side builds happen here, not inside a capture
    # lean step (tg_step_io.lean): the benchmark reads neither the involved set nor its size, so a direct-form eager step
    # does not form it (the library ignores the flag everywhere else, e.g. with the in-step lazy restart of C3)
    direct = eager and os.environ.get('TG_EAGER_DIRECT', '1') != '0' and not args.eager_copy  # no compact copy
    lean = direct and not args.no_lean  # (with the in-step restart loop of C3: the flags are marked, no sorted set is formed)
    buf.io.lean = 1 if lean else 0

    # ---- untimed: state pre-roll, then the contract's warm-up steps.  All eager launches but the last two warm-up
    # steps, which are replays of the graph the timed region replays (the first replay of a fresh graph pays its upload)
    n_untimed = preroll + warmup
    # steps per captured graph: consecutive replays of a graph are ~9 us apart on the device (tools/rocpd_gaps.py: the kernels
    # inside a replay follow each other within 0-2 us, the first kernel of the next replay starts 8.7 us after the last of
    # this one), so the timed region replays graphs of several steps - the step reads its batch at a device-side offset and
    # advances it, a graph of g steps is g copies of the same five launches.  g: the largest divisor of K up to 25; one
    # untimed replay uploads the graph (its g batches are the last of the untimed pre-roll + warm-up batches)
    gsteps = 1
    if not args.no_graph and n_untimed >= 6:
        for gcand in range(min(25, steps, n_untimed - 4), 0, -1):
            if steps % gcand == 0:
                gsteps = gcand
                break
        if args.graph_steps:
            gsteps = max(1, min(args.graph_steps, steps, n_untimed - 4))
            while steps % gsteps:
                gsteps -= 1
    n_replay_warm = 0 if (args.no_graph or n_untimed < 6) else (gsteps if gsteps > 1 else min(2, warmup))
    u_trace = []
    for b in range(n_untimed - n_replay_warm):
        full_form = lean and b in (n_untimed // 2, n_untimed - n_replay_warm - 2)  # two full steps: the involved set's size
        if full_form:
            buf.io.lean = 0
        model.launch_step(buf)
        if full_form:
            buf.io.lean = 1
            u_trace.append((b, int(buf.counts[0].item())))
    torch.cuda.synchronize()
    assert int(buf.err.item()) == 0, f'invariant word {int(buf.err.item())}'
    cnt = buf.counts.tolist()
    model.note_rows(cnt[1], cnt[2])  # bounds on the updater's rows (steady state reached): pending messages / unique positives
    assert (cnt[0] == -1) == lean, (cnt, lean)  # the lean form was taken exactly where expected

    # ---- timed region: K steps, hipGraph replay of one captured step
    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        snap = [t.clone() for t in (buf.offset,) + ((buf.lazy_batch,) if restart_prob > 0 else ())]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(gsteps):
                model.launch_step(buf)
        buf.offset.copy_(snap[0])  # capture does not execute: offset unchanged; make sure
        if restart_prob > 0:
            buf.lazy_batch.copy_(snap[1])
        for _ in range(n_replay_warm // gsteps):
            graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // gsteps if graph is not None else steps):
        if graph is not None:
            graph.replay()
        else:
            model.launch_step(buf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(buf.err.item()) == 0, f'invariant word {int(buf.err.item())}'
    assert int(buf.offset.item()) == (preroll + warmup + steps) * B
    rep_ms = []
    for _ in range(reps if graph is not None else 0):
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        for _ in range(steps // gsteps):
            graph.replay()
        torch.cuda.synchronize()
        rep_ms.append((time.perf_counter() - r0) / steps * 1e3)
    assert int(buf.offset.item()) == (preroll + warmup + steps * (1 + len(rep_ms))) * B
    self_check = None
    if graph is not None and not args.no_self_check and stream['n_nodes'] <= 2_000_000:
        self_check = replay_self_check(cfg, args, stream, resident, model, buf, n_untimed - n_replay_warm,
                                       steps * (1 + len(rep_ms)) + n_replay_warm, cnt, trig if restart_prob > 0 else None, lean)

    # ---- per-stage timing on the next unseen batches, live (HIP events on the launch stream)
    names, stage_ms, counts = profile_stages(model, buf, n_prof)
    assert int(buf.err.item()) == 0, f'invariant word {int(buf.err.item())} after the profiling pass'
    U, O_, P = counts[0], counts[1], counts[2]
    if lean:  # the sets were not formed in the timed form: their sizes (for the algorithmic byte counts) from two full steps
        buf.io.lean = 0
        _, _, cf = profile_stages(model, buf, 2, record=False)
        buf.io.lean = 1
        U, O_ = cf[0], cf[1]
    from www2023tiger_amd._lib import lib as _tg
    tile = bool(fused and _tg.tg_attn_tile_applies(C.byref(model.model_struct())))
    # eager query rows (tg_model.g_table) in use; with the in-step restart loop only in a lean step with the centre-row table
    gtab = getattr(model, '_gtab', None) is not None and (restart_prob == 0 or (lean and getattr(model, '_ctab', None) is not None))
    work = stage_work(cfg, U, O_, P, eager, fused, stream['n_nodes'], E, tile, gtab)
    traffic = load_traffic(traffic_tag)
    empty = {'zero_flags', 'dedup_positive', 'restarter_targets', 'apply_messages(gru)' if eager else 'eager_updater(gru)'}
    if lean:  # no compaction launch, and the centres ride on the sampler's launch
        empty |= {'unique_compact'} | ({'attn_centres+qconst'} if fused else set())
    if direct:  # ... and STEP 4-6 are one launch (reported under writeback_phase1)
        empty |= {'gather_right_memory', 'writeback_phase0'}
        w0, w1 = work['writeback_phase0'], work['writeback_phase1']
        work['writeback_phase1'] = (None, w0[1] + w1[1], 'tg::k_writeback_fused')
    if fused:
        empty |= {'attn_gemm_g', 'attn_gemm_v', 'attn_gemm_out'}
    if tile:
        empty |= {'attn_gemm_q', 'attn_gemm_fc1', 'attn_gemm_fc2'}
    if gtab:
        empty |= {'attn_gemm_q'}
    else:
        empty |= {'eager_query_rows(G)'}
    # riders: where they apply the step has no write-back launch (STEP 4-5 ride on fc2's launch, STEP 6's rows
    # leave its epilogue) and no collate launch (sampler + centres of the NEXT batch ride on the query-row product's)
    # (B > 16 384: the same two pieces as launches on the library's side stream - write-back beside fc1, the next batch's
    # sampler beside the updater: csrc/tg_model.hip, SideLane; opt-in with TG_SIDE_STREAM=1, measured not faster)
    side = B > 16384 and os.environ.get('TG_SIDE_STREAM', '0') != '0'
    wb_rides = direct and fused and (B <= 16384 or side) and os.environ.get('TG_WB_RIDER', '1') != '0'
    prefetch = bool(buf.io.prefetch_state) and lean and gtab and (B <= 16384 or side) and os.environ.get('TG_PREFETCH', '1') != '0'
    if wb_rides and stage_ms[names.index('writeback_phase1')] < 0.5 * stage_ms[names.index('attn_gemm_fc2')]:
        empty |= {'writeback_phase1'}
    else:
        wb_rides = False
    if prefetch and stage_ms[names.index('sample_recent_edges')] < 0.5 * stage_ms[names.index('eager_query_rows(G)')]:
        empty |= {'sample_recent_edges'}
    else:
        prefetch = False
    overhead = float(np.median([v for n, v in zip(names, stage_ms) if n in empty]))  # cost of an empty event pair
    stages = {n: float(v) for n, v in zip(names, stage_ms) if n not in empty}
    dom = max(stages, key=stages.get)
    all_ms = [dt / steps * 1e3] + rep_ms
    tables = {k: (int(t.numel()) * 4 if t is not None else 0) for k, t in
              (('g_table', getattr(model, '_gtab', None)), ('c_table', getattr(model, '_ctab', None)),
               ('pending_vals', model._pending))}
    out = dict(value=steps * B / dt, ms_per_step=dt / steps * 1e3,
               timed_region=dict(steps=steps, graph_replays=(steps // gsteps if graph is not None else 0),
                                 repeats_after=len(rep_ms), ms_per_step_all=[round(v, 5) for v in all_ms],
                                 ms_per_step_min=round(min(all_ms), 5), ms_per_step_max=round(max(all_ms), 5),
                                 note='value / ms_per_step are the FIRST K steps (the contract\'s timed region); the repeats '
                                      'are the same K-step replay on the following batches of the stream'),
               config=dict(derived_tables_bytes=dict(tables, total=sum(tables.values()),
                                                     note='per GPU, on top of the reference\'s state (memories, mailbox): '
                                                          'query rows, centre rows, eager-update rows'),
                           workload=cfg['name'], batch=B, dim=d, n_neighbors=K, msg_src=cfg['msg_src'],
                           upd_src=cfg['upd_src'], n_nodes=stream['n_nodes'], events=E, mode='stream (no_grad) STEP 1-6',
                           launch=(f'hipGraph replay, {gsteps} step{"s" if gsteps > 1 else ""} per captured graph' if graph is not None else 'eager'),
                           attention_weights=('pre-multiplied (tg_attn_fuse)' + (', whole block in one launch with G / S in LDS (k_attn_tile)' if tile else '')) if fused else 'as stored',
                           updater=('eager: once per stored message (TIGE.eager_updates)' + (', rows read from the tables directly' if direct else ', compact reprs copy')) if eager else
                                   'lazy: on the fly for every involved node with a pending message',
                           query_rows=('eager: per-node table of folded queries, refreshed for the batch\'s positive nodes at the end of '
                                       'the step (tg_model.g_table)' if gtab else 'G product over the 3B centres of the batch'),
                           involved_set=('not formed (tg_step_io.lean: nothing in a direct-form eager step reads it)' if lean else 'formed (sorted unique ids + ranks)'),
                           write_back=(('STEP 4-5 as a launch on the library\'s side stream beside fc1 (a parallel branch of the captured graph), '
                                        'STEP 6 rows from the epilogue of fc2' if side else
                                        'rides on the launch of fc1 or fc2 (STEP 4-5 as extra workgroups, STEP 6 rows from the epilogue of '
                                        'fc2): the stage times of the two products include it') if wb_rides else 'own launch'),
                           collate=(('sampler of the NEXT batch on the side stream beside the updater and the query rows, its centres pass '
                                     'behind them (tg_step_io.prefetch_state); every replay runs exactly one collate' if side else
                                     'sampler + centres of the NEXT batch ride on the query-row launch (tg_step_io.prefetch_state): '
                                     'stage eager_query_rows(G) includes them; every replay runs exactly one collate') if prefetch else 'first launch of the step'),
                           state_preroll_batches=preroll, involved_per_batch=float(U),
                           involved_start_end=dict(before_timed_region=(u_trace[-1][1] if u_trace else None), after=float(U),
                                                   growth=(float(U) / u_trace[-1][1] - 1.0) if u_trace and u_trace[-1][1] > 0 else None,
                                                   note='involved nodes per batch just before the timed region (a full-form '
                                                        'step of the pre-roll) and after it (the stage pass)'),
                           involved_before_timed_region=[dict(batch=b, involved=u) for b, u in u_trace],
                           outdated_per_batch=float(O_),
                           unique_pos_per_batch=float(P)),
               roofline=roofline_of(dom, stages[dom], work, traffic, overhead),
               stages_ms={n: round(v, 5) for n, v in stages.items()}, stage_event_overhead_ms=round(overhead, 5),
               kernels_ms={k: dict(avg_ms=round(float(v[0]), 5), launched=v[1]) for k, v in KERNEL_MS.items()},
               kernels_ms_note='kernel-bound HIP events of the step\'s main launches in the per-stage pass (eager launches; the '
                               'dispatch\'s own begin / end timestamps); stages_ms are event INTERVALS around the same '
                               'launches: each includes its two event records and, for a timed kernel, the bound pair')
    if self_check is not None:
        out['replay_self_check'] = self_check
    if restart_prob > 0:
        out['config'].update(restart_prob=restart_prob, restart_triggers_in_timed_region=n_trig,
                             restarter='static, re-initialisation inside the step (tg_lazy_restart)')
    # the memory-gather kernel (north_star: ">= 40 % of HBM roofline for the memory-gather kernel"): the launch of the TIMED
    # step that gathers the involved nodes' memory rows (STEP 1-2, tiger.py:214-221).  Eager updates, direct form: that is
    # the attention core - it reads the rows from the per-node tables itself, there is no copy launch.  It is priced on its
    # COMPULSORY bytes (every involved node's row once + the edge rows + the neighbour lists: what no design could avoid),
    # timed by the kernel-bound events of this run; the same duration priced on the design's bytes (+ the per-centre G-row
    # in / S-row out streams) and on the PMC bytes is beside it.  Otherwise (copy form / lazy form): the gather into reprs.
    g_name, u_name = 'attn_core(gather+softmax)' if direct else 'gather_right_memory', \
        'eager_updater(gru)' if eager else 'apply_messages(gru)'
    mg = roofline_of(g_name, stages[g_name], work, traffic, overhead)
    fe = 0 if cfg.get('no_feats') else 1
    nk_ = 2 * (2 * d + (d if (fe or not fused) else 0))
    Qn = 3 * B
    byts = dict(unique_node_rows=float(U * 4 * d * (1 + fe)), edge_rows=float(Qn * K * 4 * d * fe),
                neighbour_lists=float(Qn * K * 20), g_s_streams=float(2.0 * Qn * nk_ * 4))
    byts['compulsory'] = byts['unique_node_rows'] + byts['edge_rows'] + byts['neighbour_lists']
    byts['design'] = byts['compulsory'] + byts['g_s_streams']
    if direct:
        core = mg
        t_s = core['avg_ms'] * 1e-3
        comp_gbs = byts['compulsory'] / t_s / 1e9
        pmc = core.get('traffic')
        mg = dict(bound='hbm', kernel=g_name, device_kernel=core['device_kernel'], launched=core.get('launched'),
                  avg_ms=core['avg_ms'], timing=core['timing'], avg_ms_event_interval=core['avg_ms_event_interval'],
                  rocprof_avg_ms_stored=core.get('rocprof_avg_ms_stored'),
                  algorithmic_bytes=byts['compulsory'], achieved=comp_gbs, peak=HBM_PEAK_GBS, unit='GB/s',
                  frac=comp_gbs / HBM_PEAK_GBS, traffic=pmc, traffic_kernel=core.get('traffic_kernel'),
                  frac_by_design_bytes=byts['design'] / t_s / 1e9 / HBM_PEAK_GBS,
                  frac_by_pmc_bytes=(pmc / t_s / 1e9 / HBM_PEAK_GBS) if pmc else None,
                  bytes=dict(byts, pmc=pmc),
                  north_star_40pct_of_hbm_met=bool(comp_gbs / HBM_PEAK_GBS >= 0.40),
                  note='the kernel the timed step launches (k_attn_core gathers the involved rows itself); frac = COMPULSORY bytes '
                       '/ its kernel-bound duration in THIS run; rocprof_avg_ms_stored is the rocprofv3 average of a builder-run '
                       'session of the same command and pre-roll (profiles/), i.e. another box and other batches - its involved '
                       'set differs by a few per cent; the stand-alone copy-form gather is a footnote (copy_form_gather)')
        if not getattr(args, 'no_copy_form', False):
            # footnote: the stand-alone gather launch of the COPY form of the same step (tg_step_io.eager_copy: reprs[u] =
            # pending-or-right row), which the timed step does not issue
            buf.io.eager_copy = 1
            n2, st2, c2 = profile_stages(model, buf, 4, record=False)
            buf.io.eager_copy = 0
            assert int(buf.err.item()) == 0
            U2, O2 = c2[0], c2[1]
            t2 = float(st2[n2.index('gather_right_memory')])
            b2 = 2.0 * 4 * d * U2 + 12 * O2
            mg['copy_form_gather'] = dict(
                kernel='gather_right_memory', device_kernel='tg::k_consume_gather_check<true>', avg_ms_event_interval=t2,
                algorithmic_bytes=float(b2), gbs=b2 / (t2 * 1e-3) / 1e9, frac=b2 / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                traffic=kernel_traffic(load_traffic(traffic_tag + '_copy'), 'tg::k_consume_gather_check'),
                note='NOT part of the timed step: the copy form (tg_step_io.eager_copy = 1) run on 4 further batches')
    mw = 4 * d
    not_right = 0 if cfg['upd_src'] == 'right' else 1
    survey_bytes = U * 4 * d + O_ * (4 * mw + 4) + O_ * (4 * d + 4) * not_right + U * 4 * d
    t_gather = mg['avg_ms'] if direct else stages[g_name]  # direct: the core's kernel-bound duration
    t_both = (t_gather + stages[u_name]) * 1e-3
    mg['survey_bytes_gather'] = dict(
        formula='U*4d + O*(16d+4) + O*(4d+4)*[upd_src != right] + U*4d  (SURVEY.md s8 d)', bytes=float(survey_bytes),
        kernels=[g_name, u_name], ms=float(t_gather + stages[u_name]),
        gbs=survey_bytes / t_both / 1e9, frac=survey_bytes / t_both / 1e9 / HBM_PEAK_GBS,
        note='the mailbox / updater-source rows are gathered inside the updater launch, which is MFMA-bound'
             + (' and, with eager updates, runs on the P nodes that received a message instead of the O nodes that hold one' if eager else ''))
    out['roofline_memory_gather'] = mg
    for r in (out['roofline'], mg):
        r['traffic_source'] = TRAFFIC_SOURCE.get(traffic_tag) if r.get('traffic') is not None else None
    out['roofline_updater'] = roofline_of(u_name, stages[u_name], work, traffic, overhead)
    out['roofline_neighbour_gather'] = roofline_of('attn_core(gather+softmax)', stages['attn_core(gather+softmax)'], work, traffic, overhead)
    r = out['roofline_neighbour_gather']
    t_s = r['avg_ms'] * 1e-3
    r['bytes_three_ways'] = dict(byts, pmc=r.get('traffic'),
                                 frac_of_hbm_peak=dict(compulsory=byts['compulsory'] / t_s / 1e9 / HBM_PEAK_GBS,
                                                       design=byts['design'] / t_s / 1e9 / HBM_PEAK_GBS,
                                                       pmc=(r['traffic'] / t_s / 1e9 / HBM_PEAK_GBS) if r.get('traffic') else None),
                                 note='frac / achieved of this object are priced with the DESIGN bytes')
    # the sampler on the HBM roofline (SURVEY.md s8 d: bytes_samp = Q (8 ceil(log2(deg + 1)) + 28 K), + the batch arrays); where
    # the collate part has a launch of its own (large batches, or no prefetch) its kernel-bound duration prices it, else it
    # rides on the query-row product's launch and has no duration of its own
    sw_ = work['sample_recent_edges']
    kb = KERNEL_MS.get(SLOT_OF_STAGE['sample_recent_edges'])
    rs = dict(bound='hbm', kernel='sample_recent_edges', algorithmic_bytes=float(sw_[1]), peak=HBM_PEAK_GBS, unit='GB/s',
              formula='Q (8 ceil(log2(deg + 1)) + 28 K) + 40 B  (SURVEY.md s8 d bytes_samp, streaming step: no history queries)')
    if 'sample_recent_edges' in stages and kb:
        rs.update(device_kernel=kb[1], avg_ms=float(kb[0]), timing='kernel-bound HIP events', achieved=sw_[1] / (kb[0] * 1e-3) / 1e9,
                  frac=sw_[1] / (kb[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=kernel_traffic(traffic, sw_[2], kb[1]),
                  note='the launch also carries the centres pass (per-centre invariants, first dedup pass, snapshot of the 2B '
                       'positive rows: ' + str(int(Qn * 4 * d * (2 + fe))) + ' more bytes) - a chain of dependent round trips per query '
                       '(indptr -> (G + 1)-ary search of the f64 times -> K-entry tail), latency-bound, not bandwidth-bound')
    else:
        rs.update(achieved=None, frac=None, avg_ms=None,
                  note='no launch of its own in the timed step: the NEXT batch\'s sampler + centres ride on the query-row launch '
                       '(tg_step_io.prefetch_state); its duration is inside stage eager_query_rows(G)')
    out['roofline_sampler'] = rs
    if want_cpu:
        out['cpu_baseline'] = cpu_baseline(stream, cfg, model)
    del buf, graph, model, resident
    torch.cuda.empty_cache()
    return out


def replay_self_check(cfg, args, stream, resident, model, buf, n_before, n_replayed, cnt, trig, lean):
    """Untimed: the batches the timed region has just replayed from ONE captured graph go through a SECOND model (same
    seed, same weights, same switches) as plain eager launches, in the same order; memories, mailbox, has-message set
    and the last batch's embeddings of the two must be equal.  The graph replay (device-side offset, baked launch
    parameters) is thereby checked against the launch form the parity tests compare with the oracle."""
    model2, _ = build_models(stream, cfg['d'], cfg['K'], cfg['msg_src'], cfg['upd_src'], restarter='static', device='cuda:0',
                             zero_nfeats=not cfg.get('no_feats'))
    if not args.no_fuse:
        model2.fuse_attention()
    if not args.no_eager:
        model2.eager_updates()
    buf2 = model2.StepBuffers(model2, cfg['B'], False, resident=resident)
    buf2.io.eager_copy = buf.io.eager_copy
    buf2.io.lean = 1 if lean else 0
    if trig is not None:
        with torch.no_grad():
            model2.restarter_fn.left_emb.weight.copy_(model.restarter_fn.left_emb.weight)
            model2.restarter_fn.right_emb.weight.copy_(model.restarter_fn.right_emb.weight)
        buf2.enable_lazy_restart(model2, trig)
    for _ in range(n_before):  # what the timed model launched eagerly before its graph was captured
        model2.launch_step(buf2)
    torch.cuda.synchronize()
    model2.note_rows(cnt[1], cnt[2])  # the same row bounds as the captured graph has baked in: the same updater blocks
    for _ in range(n_replayed):  # the replays: the last warm-up steps and the timed region
        model2.launch_step(buf2)
    torch.cuda.synchronize()
    assert int(buf2.err.item()) == 0 and int(buf2.offset.item()) == int(buf.offset.item())
    worst = 0.0
    pairs = [('left_memory.vals', model.left_memory.vals, model2.left_memory.vals),
             ('right_memory.vals', model.right_memory.vals, model2.right_memory.vals),
             ('left_memory.update_ts', model.left_memory.update_ts, model2.left_memory.update_ts),
             ('right_memory.update_ts', model.right_memory.update_ts, model2.right_memory.update_ts),
             ('h (last batch)', buf.h, buf2.h)]
    has, has2 = model.msg_store.has_msg_mask(), model2.msg_store.has_msg_mask()
    assert torch.equal(has, has2), 'self-check: has-message sets differ between graph replay and eager launches'
    idx = torch.nonzero(has).flatten()
    pairs.append(('mailbox rows', model.msg_store.node_msg_vals[idx], model2.msg_store.node_msg_vals[idx]))
    pairs.append(('mailbox ts', model.msg_store.node_msg_ts[idx], model2.msg_store.node_msg_ts[idx]))
    for name, a, b in pairs:
        diff = float((a - b).abs().max()) if a.numel() else 0.0
        assert diff == 0.0, f'self-check: {name} differs between graph replay and eager launches by {diff} (the step is bit-reproducible)'
        worst = max(worst, diff)
    del buf2, model2
    return dict(compared='memories, update times, mailbox rows / times, has-message set, last embeddings: hipGraph replay '
                         'of the timed region vs a second model driven by eager launches over the same batches',
                batches=n_before + n_replayed, replayed=n_replayed, max_abs_diff=worst)


def reference_api_loop(cfg, batch_sizes=(200, 1024), n_batches=150):
    """What INTEGRATION.md Option A delivers: the reference's own evaluation harness - the Python loop of
    tiger/eval_utils.py:15-68 (`eval_edge_prediction`: DataLoader -> collator -> contrast_learning -> scores -> AP / AUC)
    - on this package's drop-in classes, at the reference's evaluation batch size (200) and at C2's (1024).  Not the
    headline metric: the loop pays Python, the collator's host side and one call per batch."""
    from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, InteractionData
    from www2023tiger_amd.eval_utils import eval_edge_prediction
    out = {}
    n = n_batches * max(batch_sizes)
    E = max(cfg['E'], n)
    st = make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=cfg['d'])
    model, _ = build_models(st, cfg['d'], cfg['K'], cfg['msg_src'], cfg['upd_src'], restarter='static', dropout=0.1)
    model.eval()
    dev = model.device
    coll = GraphCollator(model.graph, cfg['K'], 1, restarter='static', hist_len=1)
    rs = np.random.RandomState(1)
    for bs in batch_sizes:
        m = n_batches * bs
        ev = InteractionData(st['src'][:m], st['dst'][:m], st['ts'][:m], st['eids'][:m], np.zeros(m, dtype=np.int64), seed=0,
                             eval=True, neg_dst=rs.randint(cfg['n_u'] + 1, cfg['n_u'] + cfg['n_i'] + 1, m))
        dl = BatchLoader(ev, bs, coll)
        res = {}
        for form, env in (('resident', '1'), ('per_batch_loop', '0')):
            # resident (default): the harness recognises the BatchLoader and streams the pass - columns uploaded once, one
            # call per batch, eager updates + pre-multiplied weights for the pass; per_batch_loop (TG_EVAL_RESIDENT=0): the
            # reference's loop literally - collate object, contrast_learning, clones, sigmoid per batch
            os.environ['TG_EVAL_RESIDENT'] = env
            times = []
            for _ in range(3):  # first pass warms allocations up; best of the next two
                model.reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ap_, auc_ = eval_edge_prediction(model, dl, dev, restart_mode=False)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            best = min(times[1:])
            res[form] = dict(value=m / best, unit='events/s', ms_per_batch=best / n_batches * 1e3, batches=n_batches, ap=ap_,
                             auc=auc_)
        os.environ.pop('TG_EVAL_RESIDENT', None)
        out[f'bs{bs}'] = dict(res['resident'], per_batch_loop=res['per_batch_loop'])
    # restart_mode=True - what the reference's default recipe runs (--restart_prob 0.01, --restarter_type seq --hist_len 40;
    # train_self_supervised.py:50,194-200): the lazy restart of eval_utils.py:37-42 before every batch, at the reference's
    # evaluation batch size
    del model
    torch.cuda.empty_cache()
    rm = {}
    for rst, hl in (('static', 1), ('seq', 40)):
        model, _ = build_models(st, cfg['d'], cfg['K'], cfg['msg_src'], cfg['upd_src'], restarter=rst, hist_len=hl, dropout=0.1)
        model.eval()
        coll = GraphCollator(model.graph, cfg['K'], 1, restarter=rst, hist_len=hl)
        bs, nbr = 200, 100
        m = nbr * bs
        # the LAST m events of the stream, as a validation split follows its training split: the nodes a batch restarts
        # have histories (evaluated from the stream's first event on, every restart would meet an empty history - one
        # compact row per node, the restarter at its cheapest)
        lo = len(st['src']) - m
        ev = InteractionData(st['src'][lo:], st['dst'][lo:], st['ts'][lo:], st['eids'][lo:], np.zeros(m, dtype=np.int64), seed=0,
                             eval=True, neg_dst=rs.randint(cfg['n_u'] + 1, cfg['n_u'] + cfg['n_i'] + 1, m))
        dl = BatchLoader(ev, bs, coll)
        res = {}
        for form, env in (('resident', '1'), ('per_batch_loop', '0')):
            os.environ['TG_EVAL_RESIDENT'] = env
            times = []
            for _ in range(2):
                model.reset()
                up = set()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ap_, auc_ = eval_edge_prediction(model, dl, model.device, restart_mode=True, uptodate_nodes=up)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            res[form] = dict(value=m / times[-1], unit='events/s', ms_per_batch=times[-1] / nbr * 1e3, batches=nbr, ap=ap_, auc=auc_,
                             restarted_nodes=len(up), events=f'[{lo}, {lo + m}) of the stream')
        os.environ.pop('TG_EVAL_RESIDENT', None)
        rm[f'{rst}_bs{bs}'] = dict(res['resident'], per_batch_loop=res['per_batch_loop'])
        del model
        torch.cuda.empty_cache()
    out['restart_mode'] = rm
    model = None
    out['what'] = ('www2023tiger_amd.eval_utils.eval_edge_prediction (the harness of the reference\'s tiger/eval_utils.py:15-68, same '
                   'signature) over a BatchLoader on the drop-in TIGER, whole call timed (uploads, table builds, AP / AUC included): '
                   'the resident form the harness takes by itself, and the literal per-batch loop beside it; restart_mode: '
                   'the same with restart_mode=True (lazy restart before every batch, static and seq restarter)')
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (this process has made no GPU
    call), relay rank 0's JSON line (the children inherit stdout) and exit with the launcher's code."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def restarter_roofline(stream, cfg, args, first_batch, n_batches, ms_per_step):
    """--train --train-restarter seq: the SeqRestarter's kernels on their rooflines.  The dominant product - the Q / K projection
    of the COMPACT history rows (csrc/tg_restart.hip: one row per real event, one per node for its padded slots, one shared
    last row; K = d_e + d on a zero node-feature table) - priced with the rows of the timed batches counted from the
    stream on the host and the rocprofv3 average of its launch from the committed profile of this command (a builder-run
    measurement, named as such: the training step is not instrumented per kernel); the other restarter kernels with
    their stored durations and matrix-pipe occupancy (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES) beside it."""
    import csv
    import glob
    B, d, H = cfg['B'], cfg['d'], args.hist_len
    dm, wx = 5 * d, 2 * d
    src, dst = stream['src'], stream['dst']
    rows = []
    for b in np.linspace(first_batch, first_batch + n_batches - 1, 6).astype(int):
        lo = int(b) * B
        deg = np.bincount(np.concatenate([src[:lo], dst[:lo]]), minlength=stream['n_nodes'])  # events before the batch
        pos = np.unique(np.concatenate([src[lo:lo + B], dst[lo:lo + B]]))
        dv = deg[pos]
        rows.append(1 + int(np.minimum(dv, H - 1).sum() + (dv < H - 1).sum()))
    R = float(np.mean(rows))
    out = dict(bound='mfma', kernel='q/k projection of the compact history rows (forward)', compact_rows_per_step=R,
               slots_per_step=float(len(pos) * H), algorithmic_flops=2.0 * R * wx * 2 * dm,
               note_rows='rows counted on the host from the stream (events before the batch as the history length); the slot '
                         'grid of the reference is unique positive nodes x hist_len')
    stats = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r05_train_c2_seq_kernel_stats_v*.csv')))
    stats = [f for f in stats if 'before' not in f]
    if stats:
        per = {}
        with open(stats[-1]) as f:
            for r in csv.DictReader(f):
                per[r['Name']] = (float(r['AverageNs']) * 1e-6, int(r['Calls']))
        calls = max((c for n, (t, c) in per.items() if 'k_seq_build_c' in n), default=0)
        qk = [(n, t) for n, (t, c) in per.items() if 'k_gemm_rb<2, 1>' in n]
        if qk:
            t = qk[0][1]
            ach = out['algorithmic_flops'] / (t * 1e-3) / 1e12
            out.update(device_kernel=norm_kernel(qk[0][0].split('(')[0].replace('void ', '')), avg_ms=t,
                       timing='rocprofv3 --kernel-trace average, stored (' + os.path.relpath(stats[-1], ROOT) + ')',
                       achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / MFMA_F32_PEAK_TFLOPS)
        out['restarter_kernels_ms_stored'] = {
            norm_kernel(n.split('(')[0].replace('void ', '')): round(t * c / max(calls, 1), 5)
            for n, (t, c) in per.items() if 'k_seq_' in n or 'k_gemm_tn_group' in n or 'k_gemm_rb<2, 1>' in n}
        out['restarter_kernels_ms_stored_note'] = 'per training step (average x calls per step); k_gemm_tn_group includes the contrast half\'s group'
    mf = os.path.join(ROOT, 'profiles', 'r05_pmc_mfma_train_seq.json')
    if os.path.exists(mf):
        j = json.load(open(mf))
        out['mfma_busy_stored'] = {norm_kernel(k): v['mfma_busy_frac'] for k, v in j.items()
                                   if any(x in k for x in ('k_seq_scores', 'k_gemm_rb<2, 1>', 'k_gemm_tn_group'))}
    out['ms_per_step'] = ms_per_step
    return out


def train_main(args, cfg):
    """--train: the training iteration of train_self_supervised.py:143-171 (contrast loss only,
    restart_prob == 0) as tg_train_step + tg_adam_step on the resident stream.  Not the headline
    metric; reported next to it in DESIGN.md."""
    from www2023tiger_amd.model.training import FusedTrainer
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    B, K, d = cfg['B'], cfg['K'], cfg['d']
    E = max(cfg['E'], (args.warmup + args.steps + 4) * B)
    no_feats = bool(cfg.get('no_feats'))
    stream = make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=d,
                         integer_ts=cfg.get('integer_ts', True), with_efeats=not no_feats)
    rkind = args.train_restarter
    model, _ = build_models(stream, d, K, cfg['msg_src'], cfg['upd_src'], restarter='seq' if rkind == 'seq' else 'static',
                            hist_len=args.hist_len, device='cuda:0', zero_nfeats=not no_feats, dropout=args.train_dropout)
    model.train()
    resident = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
    tr = FusedTrainer(model, B, lr=1e-4, resident=resident, mutual=rkind != 'none', mutual_coef=1.0)
    _ = model.graph.tcsr, model.model_struct()  # lazy device-side builds happen here, not inside a capture (--warmup 0)
    lazy = None
    if args.train_restart_prob > 0 and rkind != 'none':
        # the lazy-restart loop of train_self_supervised.py:152-163 in front of every iteration (FusedTrainer.enable_lazy_restart):
        # the draws are made up front; one trigger is placed in the warm-up so that the timed iterations run in the state an
        # epoch is in from its first trigger on (expected at batch 1 / restart_prob) - every batch re-initialises what it
        # involves and is not up to date.  Iterations are launched eagerly (the seq form reads one count back per iteration).
        n_it = args.warmup + args.steps
        trig = (np.random.RandomState(1).rand(n_it) < args.train_restart_prob).astype(np.uint8)
        trig[0] = 0
        trig[max(1, args.warmup // 2)] = 1
        tr.enable_lazy_restart(trig)
        args.no_graph = True
        lazy = dict(restart_prob=args.train_restart_prob, triggers_in_timed_region=int(trig[args.warmup:].sum()), restarted=[])
    # graphs of several iterations (consecutive graph replays are ~9 us apart on the device, see run_stream_leg); the last
    # warm-up iterations are one untimed replay of the graph (uploads it)
    gsteps = 1
    if not args.no_graph:
        for gcand in range(min(10, args.steps, args.warmup), 0, -1):
            if args.steps % gcand == 0:
                gsteps = gcand
                break
    n_warm_replay = gsteps if (not args.no_graph and gsteps > 1) else 0
    for _ in range(args.warmup - n_warm_replay):
        tr.launch()
    torch.cuda.synchronize()
    assert int(tr.buf.sb.err.item()) == 0
    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        off0 = tr.buf.sb.offset.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(gsteps):
                tr.launch()
        tr.buf.sb.offset.copy_(off0)
        if n_warm_replay:
            graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps // gsteps if graph is not None else args.steps):
        if graph is not None:
            graph.replay()
        else:
            tr.launch()
            if lazy is not None and rkind == 'seq':
                lazy['restarted'].append(tr.restarted)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(tr.buf.sb.err.item()) == 0
    assert int(tr.buf.sb.offset.item()) == (args.warmup + args.steps) * B
    loss = float(tr.buf.losses[0])
    assert np.isfinite(loss)
    extra = {}
    if lazy is not None:
        r = lazy.pop('restarted')
        if r:
            lazy.update(restarted_nodes_per_iteration_mean=float(np.mean(r)), restarted_nodes_per_iteration_max=int(max(r)))
        extra['lazy_restart_loop'] = lazy
    if rkind == 'seq' and lazy is None:
        try:
            extra['roofline_restarter'] = restarter_roofline(stream, cfg, args, args.warmup, args.steps, dt / args.steps * 1e3)
        except Exception as e:  # a side object: the line does not depend on it
            extra['roofline_restarter'] = dict(error=repr(e))
    print(json.dumps(dict(extra, metric='training interaction-events/sec (collate + STEP 1-7 + backward + Adam)',
                          value=args.steps * B / dt, unit='events/s', n_gpus=1, steps=args.steps, warmup=args.warmup,
                          ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None,
                          dtype='f32', data='synthetic',
                          config=dict(workload=cfg['name'], batch=B, dim=d, n_neighbors=K,
                                      mode='train (contrast only)' if rkind == 'none' else
                                      f'train (contrast + mutual, {rkind} restarter, hist_len {args.hist_len})',
                                      dropout=args.train_dropout, last_mutual_loss=float(tr.buf.losses[1]),
                                      launch=(f'hipGraph replay, {gsteps} iteration{"s" if gsteps > 1 else ""} per captured graph' if graph is not None else 'eager'), last_loss=loss))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS), help='c2 is the benchmarked configuration')
    ap.add_argument('--preroll', type=int, default=None,
                    help=f'untimed state pre-roll batches before --warmup (default {PREROLL}; 96 for c5s / c5)')
    ap.add_argument('--no-c5s-leg', action='store_true', help='default C2 run: skip the short HBM-roofline leg')
    ap.add_argument('--no-api-loop', action='store_true',
                    help='default C2 run: skip the reference_api_loop leg (eval_edge_prediction on the drop-in API)')
    ap.add_argument('--no-dist-leg', action='store_true',
                    help='default C2 run: skip the one-rank leg of the multi-GPU code path (partitioned_form_1rank)')
    ap.add_argument('--repeats', type=int, default=4,
                    help='further, separately timed repetitions of the K-step region (spread; not part of `value`)')
    ap.add_argument('--graph-steps', type=int, default=0,
                    help='steps per captured hipGraph (default: the largest divisor of --steps up to 25)')
    ap.add_argument('--no-prefetch', action='store_true',
                    help='do not run the next batch\'s sampler + centres as riders of the step\'s last launch')
    ap.add_argument('--no-lean', action='store_true',
                    help='form the involved / outdated sets in every step even where nothing reads them')
    ap.add_argument('--no-graph', action='store_true', help='launch steps eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-self-check', action='store_true',
                    help='skip the untimed comparison of the replayed region with a second, eagerly launched model')
    ap.add_argument('--force-dist', action='store_true', help='run the multi-GPU code path even with one rank')
    ap.add_argument('--dist-graphs', action='store_true',
                    help='multi-GPU: replay captured hipGraphs around the exchange (experimental; default eager)')
    ap.add_argument('--dist-exchange', default='ipc', choices=['ipc', 'rccl'],
                    help='partitioned layout: ipc = one library call per step whose exchanges are kernels storing into the peers\' '
                         'exported windows (one node; graphs of several steps); rccl = eager launches around two all_to_all_single')
    ap.add_argument('--dist-mode', default='partitioned', choices=['partitioned', 'replicated'],
                    help='multi-GPU state layout (www2023tiger_amd/dist.py)')
    ap.add_argument('--dist-owner', default='balanced', choices=['balanced', 'hash'],
                    help='owner table of the multi-GPU shards: balanced from the stream\'s destination histogram (default), or a '
                         'plain hash of the node id (no knowledge of the stream; the imbalance is reported)')
    ap.add_argument('--dist-full-tables', action='store_true',
                    help='multi-GPU, partitioned layout: every rank allocates full-height state tables (global row addressing) '
                         'instead of its own rows + an arena (tg_model.row_of)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='multi-GPU: weak = B events per rank per step, strong = the global batch stays B')
    ap.add_argument('--no-fuse', action='store_true', help='keep the six-product attention (no tg_attn_fuse)')
    ap.add_argument('--no-eager', action='store_true', help='updater on the fly for every involved node (no eager_updates)')
    ap.add_argument('--eager-copy', action='store_true',
                    help='eager updates with the compact copy of the involved rows (stand-alone gather launch) instead of the direct form')
    ap.add_argument('--train', action='store_true', help='measure the training iteration instead (not the headline metric)')
    ap.add_argument('--train-restarter', default='none', choices=['none', 'seq', 'static'],
                    help='--train: add the mutual-learning loss of this restarter (none = contrast_only)')
    ap.add_argument('--train-dropout', type=float, default=0.0,
                    help='--train: dropout probability of the model (the reference default is 0.1, init_utils.py; 0 keeps the '
                         'lines of earlier rounds comparable)')
    ap.add_argument('--train-restart-prob', type=float, default=0.0,
                    help='--train with a restarter: the lazy-restart loop of train_self_supervised.py:152-163 in front of every '
                         'iteration (the reference default is 0.01); 0 = the training step alone')
    ap.add_argument('--hist-len', type=int, default=40, help='--train-restarter seq: history length (reference default 40, init_utils.py:58)')
    args = ap.parse_args()
    cfg = dict(WORKLOADS[args.workload])
    if args.train:
        return train_main(args, cfg)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.gpus > 1 or world > 1 or args.force_dist:
        from www2023tiger_amd import dist as tdist
        return tdist.bench_main(args, cfg, make_stream, build_models, rank, local_rank, world)

    preroll = args.preroll if args.preroll is not None else (C5S_PREROLL if args.workload in ('c5s', 'c5') else PREROLL)
    n_prof = max(4, min(args.steps, 30 if cfg['B'] <= 8192 else 6))
    leg = run_stream_leg(cfg, args, preroll, args.warmup, args.steps, n_prof, traffic_tag=args.workload,
                         want_cpu=not args.no_cpu_baseline and args.workload == 'c2')
    out = dict(metric='processed interaction-events/sec (memory+aggregate+embed), Wikipedia d=172',
               value=leg.pop('value'), unit='events/s', n_gpus=1, steps=args.steps, warmup=args.warmup,
               ms_per_step=leg.pop('ms_per_step'), higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype='f32', data='synthetic')
    out.update(leg)
    if args.workload == 'c2' and not args.no_c5s_leg:
        # at C2 every table is cache resident (150 MB): the HBM-roofline claim for the memory-gather kernel is made
        # on the C5-shaped tables (10 M nodes, d=256, B=65536: 70 GB of state), C5S_PREROLL untimed batches and 30 timed steps; the involved set still grows slowly
        # there (its sizes before and after the timed region are in the line)
        c5 = run_stream_leg(dict(WORKLOADS['c5s']), args, C5S_PREROLL, 4, 30, 4, traffic_tag='c5s')
        out['c5s_leg'] = dict(value=c5['value'], unit='events/s', ms_per_step=c5['ms_per_step'], steps=30, warmup=C5S_PREROLL + 4,
                              config=c5['config'], roofline_memory_gather=c5['roofline_memory_gather'],
                              roofline_sampler=c5['roofline_sampler'], roofline_updater=c5['roofline_updater'],
                              roofline_neighbour_gather=c5['roofline_neighbour_gather'], stages_ms=c5['stages_ms'])
    if args.workload == 'c2' and not args.no_api_loop:
        try:
            out['reference_api_loop'] = reference_api_loop(cfg)
        except Exception as e:  # a side leg: the headline line does not depend on it
            out['reference_api_loop'] = dict(error=repr(e))
    if args.workload == 'c2' and not args.no_dist_leg:
        # the N = 1 point of the multi-GPU code path (partitioned state, hipGraph segments around two RCCL
        # all_to_all_single with one rank) next to the single-GPU graph line, so that a scaling curve whose N > 1 points
        # come from that path has an N = 1 of the same form on record
        try:
            from www2023tiger_amd import dist as tdist
            import copy
            a2 = copy.copy(args)
            a2.steps, a2.warmup, a2.preroll = min(args.steps, 100), min(args.warmup, 20), PREROLL
            fd1 = os.dup(1)
            os.dup2(2, 1)  # RCCL prints a banner on fd 1
            try:
                leg = tdist.run_dist_leg(a2, dict(cfg), make_stream, build_models, 0, 0, 1, want_cpu=False)
            finally:
                sys.stdout.flush()
                os.dup2(fd1, 1)
                os.close(fd1)
            out['partitioned_form_1rank'] = dict(value=leg['value'], unit='events/s', ms_per_step=leg['ms_per_step'],
                                                 host_enqueue_ms_per_step=leg['host_enqueue_ms_per_step_rank0'],
                                                 steps=leg['steps'], launch=leg['config']['launch'],
                                                 stages_ms=leg['stages_ms_rank0'])
        except Exception as e:  # the headline line must not depend on RCCL coming up on a one-GPU box
            out['partitioned_form_1rank'] = dict(error=repr(e))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
