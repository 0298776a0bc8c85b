#!/usr/bin/env python3
"""Headline benchmark: processed interaction-events/sec through the TIGER event-batch hot
path (temporal sampling -> mailbox consume + GRU -> temporal attention -> memory/mailbox
write-back), BASELINE.json configs[1]: JODIE-Wikipedia-shaped stream, d=172, batch=1024,
msg_src=left upd_src=left, on 1 (or N) MI355X.

  python bench.py --gpus N --steps K --warmup W

A step = one batch of B events through tg_stream_step.  The whole synthetic stream is
resident in HBM before the timed region; the timed region replays a captured hipGraph of
one step K times (the step reads its batch at a device-side offset and advances it).
Rank 0 prints ONE JSON line (see the task contract) including
  roofline     - the dominant kernel of the step, timed live with HIP events, against the
                 HBM roofline with algorithmic bytes from SURVEY.md s8(d);
  cpu_baseline - the CPU oracle (oracle/tiger_oracle.py, "port") on a bounded sample of the
                 same workload on this host's cores.
"""
import argparse
import ctypes as C
import json
import os
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
    'c3': dict(name='C3 JODIE-Reddit-shaped synthetic', n_u=10000, n_i=984, E=672447, T=2.68e6, d=172, K=10, B=4096,
               msg_src='left', upd_src='right'),
    'c4': dict(name='C4 JODIE-LastFM-shaped synthetic (no feature tables)', n_u=980, n_i=1000, E=1293103, T=1.37e8, d=100,
               K=10, B=8192, msg_src='left', upd_src='right', no_feats=True),
    # C5 scaled to one GPU-box host: 10 M nodes as in BASELINE, 4 M events (the state tables are full size:
    # 2 x 10.2 GB memories + 41 GB mailbox), d=256, B=65536 - the HBM-roofline configuration
    'c5s': dict(name='C5 synthetic 10M nodes d=256 B=65536 (stream shortened to 4M events)', n_u=9000000, n_i=1000000,
                E=4000000, T=4.0e6, d=256, K=10, B=65536, msg_src='left', upd_src='right', no_feats=True,
                integer_ts=False),
}


def make_stream(n_u, n_i, E, T, seed=0, d_e=172, integer_ts=True, with_efeats=True):
    """SURVEY.md s8(d) generator: bipartite ids (0 = padding, users 1..n_u, items after),
    Zipf(0.8) users, Zipf(1.0) items, sorted uniform timestamps (floored: duplicates occur),
    eid = 1..E, N(0,1) edge features with row 0 = 0, one pre-drawn negative per event."""
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


def cpu_baseline(stream, cfg, model, budget_s=20.0, max_batches=40, warm=3):
    """Oracle (CPU restatement, parity-pinned to the reference) on the first batches of the same
    stream: collate + STEP 1-6 per batch, all host cores via torch's intra-op threads."""
    from oracle import tiger_oracle as O
    og = O.OracleGraph(stream['src'], stream['dst'], stream['ts'], stream['eids'], max_node_id=stream['n_nodes'] - 1)
    params = {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}
    nfeats = np.zeros((stream['n_nodes'], cfg['d']), dtype=np.float32)
    orc = O.OracleTIGER(params, og, n_nodes=stream['n_nodes'], dim=cfg['d'], nfeats=nfeats, efeats=stream['efeats'],
                        n_neighbors=cfg['K'], msg_src=cfg['msg_src'], upd_src=cfg['upd_src'], restarter='static')
    B = cfg['B']
    done, t0, elapsed = 0, None, 0.0
    with torch.no_grad():
        for b in range(warm + max_batches):
            if b == warm:
                t0 = time.perf_counter()
            sl = slice(b * B, (b + 1) * B)
            a = [stream[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
            cg = O.collate(og, a[0], a[1], a[2], a[3], cfg['K'], 'static')
            orc.stream_step(*a, cg)
            if b >= warm:
                done += 1
                elapsed = time.perf_counter() - t0
                if elapsed > budget_s:
                    break
    return dict(value=done * B / elapsed, unit='events/s', cores=torch.get_num_threads(), kind='port',
                sample=f'oracle/tiger_oracle.py, first {done} batches of B={B} after {warm} warm-up '
                       f'(collate + STEP 1-6), {elapsed:.1f} s')


def profile_stages(model, buf, steps):
    """Eager steps with the library's per-stage HIP-event timer attached (same stream)."""
    from www2023tiger_amd._lib import lib
    n = lib.tg_profiler_num_stages()
    names = [lib.tg_profiler_stage_name(i).decode() for i in range(n)]
    prof = lib.tg_profiler_create()
    assert prof, 'tg_profiler_create failed'
    buf.attach_profiler(prof)
    acc = np.zeros(n)
    counts = np.zeros(4)
    ms = (C.c_float * n)()
    for _ in range(steps):
        model.launch_step(buf)
        rc = lib.tg_profiler_read(prof, ms)
        assert rc == 0
        acc += np.array(ms[:])
        counts += buf.counts.cpu().numpy()
    buf.attach_profiler(None)
    lib.tg_profiler_destroy(prof)
    return names, acc / steps, counts / steps


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
                            hist_len=args.hist_len, device='cuda:0', zero_nfeats=not no_feats, dropout=0.0)
    model.train()
    resident = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
    tr = FusedTrainer(model, B, lr=1e-4, resident=resident, mutual=rkind != 'none', mutual_coef=1.0)
    _ = model.graph.tcsr, model.model_struct()  # lazy device-side builds happen here, not inside a capture (--warmup 0)
    for _ in range(args.warmup):
        tr.launch()
    torch.cuda.synchronize()
    assert int(tr.buf.sb.err.item()) == 0
    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        off0 = tr.buf.sb.offset.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            tr.launch()
        tr.buf.sb.offset.copy_(off0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            tr.launch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(tr.buf.sb.err.item()) == 0
    assert int(tr.buf.sb.offset.item()) == (args.warmup + args.steps) * B
    loss = float(tr.buf.losses[0])
    assert np.isfinite(loss)
    print(json.dumps(dict(metric='training interaction-events/sec (collate + STEP 1-7 + backward + Adam)',
                          value=args.steps * B / dt, unit='events/s', n_gpus=1, steps=args.steps, warmup=args.warmup,
                          ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None,
                          dtype='f32', data='synthetic',
                          config=dict(workload=cfg['name'], batch=B, dim=d, n_neighbors=K,
                                      mode='train (contrast only)' if rkind == 'none' else
                                      f'train (contrast + mutual, {rkind} restarter, hist_len {args.hist_len})',
                                      last_mutual_loss=float(tr.buf.losses[1]),
                                      launch='hipGraph replay' if graph is not None else 'eager', last_loss=loss))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS), help='c2 is the benchmarked configuration')
    ap.add_argument('--no-graph', action='store_true', help='launch steps eagerly instead of replaying a hipGraph')
    ap.add_argument('--force-dist', action='store_true', help='run the multi-GPU code path even with one rank')
    ap.add_argument('--dist-graphs', action='store_true',
                    help='multi-GPU: replay captured hipGraphs around the all-gather (experimental; default eager)')
    ap.add_argument('--no-fuse', action='store_true', help='keep the six-product attention (no tg_attn_fuse)')
    ap.add_argument('--no-eager', action='store_true', help='updater on the fly for every involved node (no eager_updates)')
    ap.add_argument('--train', action='store_true', help='measure the training iteration instead (not the headline metric)')
    ap.add_argument('--train-restarter', default='none', choices=['none', 'seq', 'static'],
                    help='--train: add the mutual-learning loss of this restarter (none = contrast_only)')
    ap.add_argument('--hist-len', type=int, default=20)
    args = ap.parse_args()
    cfg = dict(WORKLOADS[args.workload])
    if args.train:
        return train_main(args, cfg)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 or world > 1 or args.force_dist:
        from www2023tiger_amd import dist as tdist
        return tdist.bench_main(args, cfg, make_stream, build_models, rank, local_rank, world)

    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    B, K, d = cfg['B'], cfg['K'], cfg['d']
    n_batches = args.warmup + args.steps + 4
    E = max(cfg['E'], n_batches * B)
    no_feats = bool(cfg.get('no_feats'))
    stream = make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=d,
                         integer_ts=cfg.get('integer_ts', True), with_efeats=not no_feats)
    model, _ = build_models(stream, d, K, cfg['msg_src'], cfg['upd_src'], restarter='static', device='cuda:0',
                            zero_nfeats=not no_feats)
    resident = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
    if not args.no_fuse:  # streaming inference, parameters fixed: pre-multiplied attention weights (tg_attn_fuse)
        model.fuse_attention()
    if not args.no_eager:  # ... and the updater run once per stored message (TIGE.eager_updates)
        model.eager_updates()
    buf = model.StepBuffers(model, B, False, resident=resident)
    _ = model.graph.tcsr, model.model_struct()  # lazy device-side builds happen here, not inside a capture (--warmup 0)

    # ---- warm-up (untimed, eager): also brings memory / mailbox to steady state
    for _ in range(args.warmup):
        model.launch_step(buf)
    torch.cuda.synchronize()
    assert int(buf.err.item()) == 0, f'invariant word {int(buf.err.item())}'
    # what the loops of the package do with the counts they read back: a bound on the pending-message rows lets
    # the updater pick blocks sized for a one-round launch.  Not for the headline workload: the choice is frozen
    # into the captured graph, C2's row count keeps growing for a hundred batches, and a bound learnt from a short
    # warm-up would size the launch for rows it soon exceeds (the library default - capacity / node count - is used)
    if args.workload != 'c2':
        model.note_rows(int(buf.counts[1].item()))

    # ---- timed region: K steps, hipGraph replay of one captured step
    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        off0 = buf.offset.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            model.launch_step(buf)
        # capture does not execute: offset unchanged; make sure
        buf.offset.copy_(off0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            model.launch_step(buf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(buf.err.item()) == 0, f'invariant word {int(buf.err.item())}'
    assert int(buf.offset.item()) == (args.warmup + args.steps) * B
    events_per_s = args.steps * B / dt

    # ---- per-stage timing of the same step, live (HIP events on the launch stream)
    buf.offset.fill_((args.warmup) * B)  # re-run a slice of the stream: costs are state-independent enough
    names, stage_ms, counts = profile_stages(model, buf, min(args.steps, 50))
    U, O_, P = counts[0], counts[1], counts[2]
    dom = int(np.argmax(stage_ms))
    # algorithmic HBM bytes per launch (SURVEY.md s8 d), with the measured U/O/P of this run
    Q = 3 * B
    d_e = d
    fe = 0 if no_feats else 1  # feature tables present?
    bytes_by_stage = {
        'gather_right_memory': U * 4 * d * 2,                                   # read rows + compact write
        'apply_messages(gru)': O_ * (4 * (3 * d + d_e) + 4) + O_ * (4 * d + 4) + O_ * 4 * d,  # mailbox + upd rows + write
        'attn_core(gather+softmax)': Q * K * 4 * (fe * d_e + fe * d + d) + Q * 2 * (2 * d + d_e) * 4 * 2,  # efeat+nfeat+reprs rows, G in, S out
        'store_events': P * (4 * (3 * d + d_e) + 4) + 2 * B * 4 * d + B * 4 * d_e,
    }
    flops_by_stage = {'apply_messages(gru)': 2.0 * O_ * 3 * d * ((3 * d + d_e) + d)}
    kernel_of_stage = {'apply_messages(gru)': 'tg::k_gru<4, 2>', 'attn_core(gather+softmax)': 'tg::k_attn_core<2, 1, 4>',
                       'gather_right_memory': 'tg::k_consume_gather_check', 'sample_recent_edges': 'tg::k_sample_batch<16>'}
    name = names[dom]
    t_s = stage_ms[dom] * 1e-3
    # fabric/HBM bytes per launch of that kernel from the committed PMC passes (profiles/r01_hbm_traffic_v13.json,
    # collected with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same command); null if not recorded
    traffic = None
    try:
        tfile = {'c2': 'r01_hbm_traffic_v13.json', 'c5s': 'r01_hbm_traffic_c5s_v6.json'}.get(args.workload)
        tj = json.load(open(os.path.join(ROOT, 'profiles', tfile))) if tfile else {'kernels': {}}
        traffic = tj['kernels'].get(kernel_of_stage.get(name, ''), {}).get('bytes_per_launch')
    except (OSError, ValueError):
        pass
    if name in flops_by_stage:  # the GRU kernel is MFMA-bound at C2 (3.7 GFLOP vs 17 MB): price it against f32 MFMA
        ach = flops_by_stage[name] / t_s / 1e12
        roof = dict(bound='mfma', kernel=name, achieved=ach, peak=157.3, unit='TFLOP/s', frac=ach / 157.3,
                    traffic=traffic, avg_ms=float(stage_ms[dom]), algorithmic_flops=float(flops_by_stage[name]),
                    algorithmic_bytes=float(bytes_by_stage[name]),
                    hbm_gbs=bytes_by_stage[name] / t_s / 1e9)
    elif name in bytes_by_stage:
        ach = bytes_by_stage[name] / t_s / 1e9
        roof = dict(bound='hbm', kernel=name, achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS,
                    traffic=traffic, avg_ms=float(stage_ms[dom]), algorithmic_bytes=float(bytes_by_stage[name]))
    else:
        roof = dict(bound='hbm', kernel=name, achieved=None, peak=HBM_PEAK_GBS, unit='GB/s', frac=None, traffic=traffic,
                    avg_ms=float(stage_ms[dom]))
    out = dict(metric='processed interaction-events/sec (memory+aggregate+embed), Wikipedia d=172',
               value=events_per_s, unit='events/s', n_gpus=1, steps=args.steps, warmup=args.warmup,
               ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype='f32', data='synthetic',
               config=dict(workload=cfg['name'], batch=B, dim=d, n_neighbors=K, msg_src=cfg['msg_src'],
                           upd_src=cfg['upd_src'], n_nodes=stream['n_nodes'], events=E, mode='stream (no_grad) STEP 1-6',
                           launch='hipGraph replay' if graph is not None else 'eager',
                           attention_weights='pre-multiplied (tg_attn_fuse)' if not args.no_fuse else 'as stored',
                           involved_per_batch=float(U), outdated_per_batch=float(O_), unique_pos_per_batch=float(P)),
               roofline=roof,
               stages_ms={n: round(float(v), 5) for n, v in zip(names, stage_ms)})
    # the HBM-bound memory-gather kernel (right-memory rows of the involved nodes), next to the dominant kernel:
    # the north star prices THIS kernel against the HBM roofline on the C5-scaled run (at C2 the state is cache resident)
    gname = 'gather_right_memory'
    if gname in names and name != gname:
        gi = names.index(gname)
        gt = stage_ms[gi] * 1e-3
        gtraffic = None
        try:
            gtraffic = tj['kernels'].get(kernel_of_stage[gname], {}).get('bytes_per_launch')
        except (NameError, KeyError):
            pass
        out['roofline_memory_gather'] = dict(bound='hbm', kernel=gname, achieved=bytes_by_stage[gname] / gt / 1e9,
                                             peak=HBM_PEAK_GBS, unit='GB/s', frac=bytes_by_stage[gname] / gt / 1e9 / HBM_PEAK_GBS,
                                             traffic=gtraffic, avg_ms=float(stage_ms[gi]),
                                             algorithmic_bytes=float(bytes_by_stage[gname]))
    if not args.no_cpu_baseline and args.workload == 'c2':
        out['cpu_baseline'] = cpu_baseline(stream, cfg, model)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
