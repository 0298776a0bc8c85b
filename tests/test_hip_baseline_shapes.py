"""GPU parity at the shapes of BASELINE.json configs[2..4] (C3, C4, C5), through the C ABI,
against the CPU oracle on the same seeded synthetic streams.  Indices bit-exact, float32 within 1e-4 under
both measures of _util.assert_close; every test reports its worst error in the session summary.

  C3  Reddit-shaped, d=172, B=4096, msg=left upd=right, static restarter with a lazy restart mid-stream
      (train_self_supervised.py:152-163, restarters.py:254-277)
  C4  LastFM-shaped, d=100, B=8192, no feature tables, timestamps up to 1.37e8 (> 2^24: the float64 T-CSR,
      graph.py:48-51)
  C5  d=256, B=65536 on the full 10 M-node id space: the HIP engine runs on the 10 000 001-row tables
      (multi-block bitmap compaction, 64-bit row offsets, k_gru XCD dealing at 1000+ row tiles) and is
      compared with the oracle run on the order-preserving compaction of the ids that occur - node ids are
      only row addresses, so results must be invariant under that renaming - plus checks that every row
      outside the touched set is still exactly zero.
"""
import numpy as np
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device('cuda', 0)


def _batch(stream, b, B):
    sl = slice(b * B, (b + 1) * B)
    return [stream[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]


def _compare_indices(buf, cg, node_map=None):
    """neighbour lists and the involved set of the fused step against the oracle's collation; node_map
    (oracle id -> engine id, order preserving, 0 -> 0) when the oracle runs on compacted ids"""
    f = (lambda a: node_map[a]) if node_map is not None else (lambda a: a)
    np.testing.assert_array_equal(buf.l1_nids.cpu().numpy(), f(cg['l1_nids']))
    np.testing.assert_array_equal(buf.l1_eids.cpu().numpy(), cg['l1_eids'])
    np.testing.assert_array_equal(buf.l1_ts.cpu().numpy(), cg['l1_ts'])
    counts = buf.counts.cpu().numpy()
    if counts[0] == -1:  # lean step (tg_step_io.lean): the involved / outdated sets were not formed
        assert counts[1] == -1
    else:
        assert counts[0] == len(cg['involved'])
        np.testing.assert_array_equal(buf.involved.cpu().numpy()[:counts[0]], f(cg['involved']))
    assert counts[2] == len(cg['rd_nids'])
    return counts


def test_c1_wikipedia_shape_b200_seq_restarter_collate():
    """BASELINE configs[0] exactly as bench.WORKLOADS['c1'] has it: Wikipedia-shaped, d = 172, B = 200, msg = left, upd =
    right, restart_prob 0 with the SEQ restarter's collation (hist_len 40: data_loader.py:82 collates the restart data
    whatever restart_prob is), 22 consecutive batches through the reference call sequence (collator -> contrast_learning)
    against the oracle - embeddings, scores, the restart data bit-exact, the final state - and the SeqRestarter's forward
    on the collated histories of three of the batches (the surrogate rows the mutual loss would read)."""
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    from www2023tiger_amd.data.data_loader import GraphCollator
    c = bench.WORKLOADS['c1']
    B, K, d, nb, H = c['B'], c['K'], c['d'], 22, 40
    assert (B, c['upd_src'], c['msg_src'], d) == (200, 'right', 'left', 172)
    E = (nb + 1) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=7, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], restarter='seq', hist_len=H, with_oracle=True)
    coll = GraphCollator(model.graph, K, 1, restarter='seq', hist_len=H)
    to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
    worst = 0.0
    for b in range(nb):
        a = _batch(stream, b, B)
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'seq', hist_len=H)
        out = coll.collate_arrays(*a)
        hcg = out[-1]
        np.testing.assert_array_equal(hcg.layers[1][0].cpu().numpy(), cg['l1_nids'])
        np.testing.assert_array_equal(hcg.layers[1][1].cpu().numpy(), cg['l1_eids'])
        np.testing.assert_array_equal(hcg.np_computation_graph_nodes, cg['involved'])
        rd = hcg.restart_data
        np.testing.assert_array_equal(rd.index.cpu().numpy(), cg['rd_index'])
        np.testing.assert_array_equal(rd.nids.cpu().numpy(), cg['rd_nids'])
        for f, k in (('hist_nids', 'rd_hist_nids'), ('anonymized_ids', 'rd_anon'), ('hist_eids', 'rd_hist_eids'),
                     ('hist_ts', 'rd_hist_ts'), ('hist_dirs', 'rd_hist_dirs')):
            np.testing.assert_array_equal(getattr(rd, f).cpu().numpy(), cg[k], err_msg=k)
        if b in (5, 12, 21):  # the restarter on the collated histories (restarters.py:85-114; zero node features: narrow form)
            hl, hr, pt = model.restarter_fn(rd.nids, rd.ts, hcg)
            rl, rr, rp = orc.restarter_forward(cg['rd_nids'], cg['rd_ts'], cg)
            worst = max(worst, *assert_close(hl.cpu().numpy(), rl.numpy(), 'surrogate h(t-)', TOL),
                        *assert_close(hr.cpu().numpy(), rr.numpy(), 'surrogate h(t+)', TOL))
            np.testing.assert_array_equal(pt.cpu().numpy().reshape(-1), np.asarray(rp).reshape(-1))
        with torch.no_grad():
            res = model.contrast_learning(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64),
                                          to(a[3], torch.float32), to(a[4], torch.int64), hcg)
        ref = orc.contrast_learning(*a, cg)
        worst = max(worst, *assert_close(res[2].cpu().numpy(), ref['pos_scores'].detach().numpy(), 'positive scores', TOL),
                    *assert_close(res[3].cpu().numpy(), ref['neg_scores'].detach().numpy(), 'negative scores', TOL))
        model._poll_train_errors()
    compare_state_with_oracle(model, orc)


@pytest.mark.parametrize('eager', [False, True], ids=['lazy', 'eager'])
def test_c3_reddit_shape_b4096_static_lazy_restart(eager):
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    from www2023tiger_amd.data.data_loader import GraphCollator
    c = bench.WORKLOADS['c3']
    B, K, d, nb = c['B'], c['K'], c['d'], 7
    E = (nb + 1) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=3, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], restarter='static', with_oracle=True)
    # the reference initialises the static restarter's tables with zeros (restarters.py:259-260); trained values
    # are what a restart is for, so the tables get non-trivial rows here (the same in the oracle)
    torch.manual_seed(11)
    with torch.no_grad():
        for nm in ('left_emb', 'right_emb'):
            tbl = getattr(model.restarter_fn, nm).weight
            tbl.normal_(0.0, 0.5)
            orc.p[f'restarter_fn.{nm}.weight'] = tbl.detach().cpu().clone()
    if eager:
        model.eager_updates()
    coll = GraphCollator(model.graph, K, 1, restarter='static')
    restarting, uptodate, n_restarted = False, set(), 0
    for b in range(nb):
        a = _batch(stream, b, B)
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        if b == 3:  # train_self_supervised.py:152-158: forget who is up to date, drop every pending message
            restarting, uptodate = True, set()
            model.msg_store.clear()
            orc.clear_msgs()
        if restarting:  # :159-163: re-initialise the involved nodes that have not been restarted yet
            involved = coll.collate_arrays(*a)[-1].np_computation_graph_nodes
            np.testing.assert_array_equal(involved, cg['involved'])
            r = np.array(sorted(set(involved.tolist()) - uptodate), dtype=np.int64)
            t0 = np.float32(a[3].min())
            model.restart(torch.from_numpy(r).to(dev()), torch.full((len(r),), float(t0), device=dev()))
            orc.restart(r, np.full(len(r), t0, dtype=np.float32))
            uptodate.update(r.tolist())
            n_restarted += len(r)
            if b == 3:
                compare_state_with_oracle(model, orc)
        buf = model.stream_step(*a)
        ref = orc.stream_step(*a, cg).numpy()
        _compare_indices(buf, cg)
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, 'h_left', TOL)
    assert n_restarted > 5000
    compare_state_with_oracle(model, orc)


@pytest.mark.parametrize('eager', [False, True, 'lean'], ids=['lazy', 'eager', 'eager-lean-tables'])
def test_c3_in_step_lazy_restart_equals_the_reference_loop(eager):
    """BASELINE configs[2] as written: the restart draws of train_self_supervised.py:152-163 are made up front
    and the loop body (forget / clear mailbox on a hit, re-initialise the involved nodes that are not up to date
    with the static restarter) runs inside tg_stream_step; the oracle runs the reference's loop on the host."""
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    c = bench.WORKLOADS['c3']
    B, K, d, nb = c['B'], c['K'], c['d'], 9
    E = (nb + 1) * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=13, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], restarter='static', with_oracle=True)
    torch.manual_seed(12)
    with torch.no_grad():
        for nm in ('left_emb', 'right_emb'):
            tbl = getattr(model.restarter_fn, nm).weight
            tbl.normal_(0.0, 0.5)
            orc.p[f'restarter_fn.{nm}.weight'] = tbl.detach().cpu().clone()
    lean = eager == 'lean'
    if eager:
        model.eager_updates()
    if lean:
        # bench.py --workload c3's form: pre-multiplied weights and the per-node query-row / centre-row tables, which the
        # restart loop's kernel keeps current itself (rows of the re-initialised nodes); the involved flags are marked,
        # no sorted set is formed
        model.fuse_attention()
    trigger = np.zeros(nb, dtype=np.uint8)
    trigger[[2, 6]] = 1          # two hits: the second one forgets the nodes restarted after the first
    buf = model.step_buffers(B, lean=lean).enable_lazy_restart(model, trigger)
    restarting, uptodate = False, set()
    for b in range(nb):
        a = _batch(stream, b, B)
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        n_r = 0
        if trigger[b] and b:
            restarting, uptodate = True, set()
            orc.clear_msgs()
        if restarting:
            r = np.array(sorted(set(cg['involved'].tolist()) - uptodate), dtype=np.int64)
            orc.restart(r, np.full(len(r), np.float32(a[3].min()), dtype=np.float32))
            uptodate.update(r.tolist())
            n_r = len(r)
        got = model.stream_step(*a, lean=lean)
        assert got is buf
        ref = orc.stream_step(*a, cg).numpy()
        counts = _compare_indices(buf, cg)
        assert counts[3] == n_r, (b, counts, n_r)
        if lean:
            assert counts[0] == -1 and model._gtab is not None and model._ctab is not None and model._gtab_stamp is not None
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, 'h_left', TOL)
        if b in (2, 3, 6):
            compare_state_with_oracle(model, orc)
    assert int(buf.lazy_batch) == nb and int(buf.lazy_restarting) == 1
    up = buf.lazy_uptodate.cpu().numpy().view(np.uint64)
    bits = np.unpackbits(up.view(np.uint8), bitorder='little')[:model.n_nodes]
    np.testing.assert_array_equal(np.nonzero(bits)[0], np.array(sorted(uptodate)))
    compare_state_with_oracle(model, orc)


def test_c4_lastfm_shape_b8192_no_feature_tables_large_timestamps():
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    c = bench.WORKLOADS['c4']
    B, K, d, nb = c['B'], c['K'], c['d'], 5
    E = (nb + 1) * B
    # T is NOT scaled down with the shortened stream: the events span the full 1.37e8 s of LastFM, so most
    # timestamps exceed 2^24 and neighbouring events collapse in float32
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'], seed=4, d_e=d, with_efeats=False)
    ts = stream['ts']
    assert (ts > 2 ** 24).mean() > 0.8
    assert (ts.astype(np.float32).astype(np.float64) != ts).mean() > 0.5   # float32 cannot hold these times
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], with_oracle=True, zero_nfeats=False)
    assert model.raw_feat_getter.nfeats is None and model.raw_feat_getter.efeats is None
    model.eager_updates()
    for b in range(nb):
        a = _batch(stream, b, B)
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        buf = model.stream_step(*a)
        ref = orc.stream_step(*a, cg).numpy()
        _compare_indices(buf, cg)
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, 'h_left', TOL)
    compare_state_with_oracle(model, orc)
    # with attention weights pre-multiplied (the benchmarked inference form) on the next batch
    model.fuse_attention()
    a = _batch(stream, nb, B)
    cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
    buf = model.stream_step(*a, lean=True)   # and as a lean step (no involved set; 2 982 node ids < 3B(K+1) slots)
    assert int(buf.counts[0]) == -1
    assert_close(buf.h[:2 * B].cpu().numpy(), orc.stream_step(*a, cg).numpy(), 'h_left (fused weights)', TOL)
    compare_state_with_oracle(model, orc)


def test_c5_d256_b65536_ten_million_node_tables():
    import bench
    from oracle import tiger_oracle as O
    c = bench.WORKLOADS['c5s']
    B, K, d, nb = c['B'], c['K'], c['d'], 3
    E = nb * B
    stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=5, d_e=d, integer_ts=False,
                               with_efeats=False)
    N = stream['n_nodes']
    assert N == 10_000_001
    model, _ = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], zero_nfeats=False)
    model.fuse_attention()
    model.eager_updates()   # the benchmarked form: updater rows precomputed per stored message
    # the oracle on the compacted id space (order preserving, padding id 0 kept)
    used = np.unique(np.concatenate([[0], stream['src'], stream['dst'], stream['neg']]))
    to_c = lambda x: np.searchsorted(used, x)
    cs = dict(stream, src=to_c(stream['src']), dst=to_c(stream['dst']), neg=to_c(stream['neg']), n_nodes=len(used))
    og = O.OracleGraph(cs['src'], cs['dst'], cs['ts'], cs['eids'], max_node_id=len(used) - 1)
    from www2023tiger_amd import hip_ops
    used_t = torch.from_numpy(used).to(dev())
    # rows of the 10 M-row tables are read with the library's own gather (64-bit row offsets); the restarter's
    # per-node tables play no part in a stream without restarts
    rows = lambda table, ids: hip_ops.gather_rows(table, ids).cpu().numpy()
    params = {k: (np.zeros((len(used), d), dtype=np.float32) if k.startswith('restarter_fn.')
                  else v.detach().cpu().numpy()) for k, v in model.named_parameters()}
    orc = O.OracleTIGER(params, og, n_nodes=len(used), dim=d, nfeats=None, efeats=None, n_neighbors=K,
                        msg_src=c['msg_src'], upd_src=c['upd_src'], restarter='static')
    for b in range(nb):
        a, ac = _batch(stream, b, B), _batch(cs, b, B)
        cg = O.collate(og, ac[0], ac[1], ac[2], ac[3], K, 'static')
        lean = b == nb - 1  # the last batch as a lean step: 10 M dedup slots indexed by node id, cleaned by position
        buf = model.stream_step(*a, lean=lean)
        ref = orc.stream_step(*ac, cg).numpy()
        counts = _compare_indices(buf, cg, node_map=used)
        assert (counts[0] == -1) == lean
        assert lean or (counts[1] == (0 if b == 0 else counts[1]) and (b == 0 or counts[1] > 1000))   # pending messages consumed
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, 'h_left', TOL)
    # state: touched rows against the oracle, every other row of the 10 M-row tables still exactly zero
    has = model.msg_store.has_msg_mask()
    np.testing.assert_array_equal(has.nonzero().flatten().cpu().numpy(), used[np.nonzero(orc.has_msg)[0]])
    def nonzeros(t):  # exact count over a table of more than 2^31 elements, in slices
        flat = t.reshape(-1)
        return sum(int(torch.count_nonzero(flat[a:a + 2 ** 28])) for a in range(0, flat.numel(), 2 ** 28))
    for nm, mem, ov, ot in (('left memory', model.left_memory, orc.left_vals, orc.left_ts),
                            ('right memory', model.right_memory, orc.right_vals, orc.right_ts)):
        got = rows(mem.vals, used_t)
        assert_close(got, ov.numpy(), nm, TOL)
        np.testing.assert_array_equal(mem.update_ts[used_t].cpu().numpy(), ot.numpy())
        assert nonzeros(mem.vals) == int(np.count_nonzero(got))          # nothing outside the touched rows
        assert nonzeros(mem.update_ts) == int(np.count_nonzero(ot.numpy()))
    hm = used[np.nonzero(orc.has_msg)[0]]
    hm_t = torch.from_numpy(hm).to(dev())
    assert_close(rows(model.msg_store.node_msg_vals, hm_t), orc.msg_vals.numpy()[np.nonzero(orc.has_msg)[0]],
                 'mailbox', TOL)
    np.testing.assert_array_equal(model.msg_store.node_msg_ts[hm_t].cpu().numpy(),
                                  orc.msg_ts.numpy()[np.nonzero(orc.has_msg)[0]])
    # ids near the top of the 10 M range took part (64-bit row offsets: row 9.99e6 * 1024 floats > 2^32 bytes)
    assert used.max() > 9_900_000 and int(has.nonzero().max()) * (3 * d + d) * 4 > 2 ** 32
