"""Parity of the HIP path (through the C ABI) with the golden vectors of the reference
and with the CPU oracle.  Needs a real MI355X: every test is marked gpu.

Bars: integer / index outputs bit-exact; float32 embeddings and memory state within
1e-4 relative - the tolerance BASELINE.json's north_star states ("within 1e-4 relative fp32
and bit-exact neighbor indices") - under BOTH measures of _util.assert_close:
max |a-b| / max(1, max|ref|) over the tensor and the per-row ||a_i-b_i|| / ||ref_i|| (rows with
||ref_i|| > 1e-3).  The worst values per test are printed in the session summary."""
import numpy as np
import pytest
import torch

from _util import (MODEL_FIXTURES, TWO_LAYER_FIXTURES, assert_close, fixture_params, fixture_tables, load, n_batches, parse_cfg, rel_err,
                   row_rel_err)

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device('cuda', 0)


def build_hip_model(z, cfg, strategy='recent_edges', dropout=0.1):
    from www2023tiger_amd.data.data_loader import GraphCollator
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.model.feature_getter import NumericalFeature
    from www2023tiger_amd.model.restarters import SeqRestarter, StaticRestarter
    from www2023tiger_amd.model.tiger import TIGER
    n_nodes, nfeats, efeats = fixture_tables(z, cfg)
    g = Graph.from_arrays(z['src'], z['dst'], z['ts'], z['eids'], strategy=strategy, seed=0, device=dev())
    assert g.num_node == n_nodes
    fg = NumericalFeature(None if nfeats is None else torch.from_numpy(nfeats),
                          None if efeats is None else torch.from_numpy(efeats), dim=cfg['d'], device=dev())
    fg.n_nodes, fg.n_edges = n_nodes, len(z['src'])
    if cfg['restarter'] == 'seq':
        rst = SeqRestarter(raw_feat_getter=fg, graph=g, hist_len=cfg['H'], n_head=2, dropout=dropout)
    else:
        rst = StaticRestarter(raw_feat_getter=fg, graph=g)
    model = TIGER(raw_feat_getter=fg, graph=g, restarter=rst, n_neighbors=cfg['K'], hit_type=cfg.get('hit', 'bin'),
                  n_layers=cfg.get('L', 1), n_head=2, dropout=dropout, msg_src=cfg['msg_src'], upd_src=cfg['upd_src'],
                  msg_tsfm_type=cfg.get('tsfm', 'id'), mem_update_type=cfg.get('upd_fn', 'gru'))
    params = fixture_params(z, cfg)
    own = dict(model.named_parameters())
    assert set(own) == set(params), set(own) ^ set(params)
    with torch.no_grad():
        for k, v in params.items():
            own[k].copy_(torch.from_numpy(v))
    model = model.to(dev()).eval()
    coll = GraphCollator(g, cfg['K'], cfg.get('L', 1), restarter=cfg['restarter'], hist_len=cfg.get('H'))
    return model, g, coll


# ------------------------------------------------------------------------------ sampler
@pytest.fixture(scope='module')
def samp():
    return load('sampler')


def _graph(z, strategy):
    from www2023tiger_amd.data.graph import Graph
    return Graph.from_arrays(z['src'], z['dst'], z['ts'], z['eids'], strategy=strategy, seed=3, device=dev())


@pytest.mark.parametrize('strategy', ['recent_edges', 'recent_nodes'])
@pytest.mark.parametrize('K', [1, 5, 10, 40])
def test_sampler_bit_exact(samp, strategy, K):
    g = _graph(samp, strategy)
    assert g.num_node == int(samp['num_node'])
    res = g.sample_temporal_neighbor(samp['q_nids'], samp['q_ts'], K)
    for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
        exp = samp[f'{strategy}_K{K}_{nm}']
        assert a.dtype == exp.dtype and a.shape == exp.shape
        np.testing.assert_array_equal(a, exp, err_msg=nm)


def test_graph_from_adj_list_matches_from_arrays(samp):
    from www2023tiger_amd.data.graph import Graph
    n = int(samp['num_node'])
    adj = [[] for _ in range(n)]
    for s, d, t, e in zip(samp['src'], samp['dst'], samp['ts'], samp['eids']):
        adj[s].append((d, e, t, 0))
        adj[d].append((s, e, t, 1))
    g = Graph(adj, strategy='recent_edges', seed=3, device=dev())
    res = g.sample_temporal_neighbor(samp['q_nids'], samp['q_ts'], 10)
    for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
        np.testing.assert_array_equal(a, samp[f'recent_edges_K10_{nm}'], err_msg=nm)


@pytest.mark.parametrize('K', [5, 10])
def test_sampler_uniform_mt19937_stream(samp, K):
    """Draws must follow numpy's legacy RandomState.randint stream (two consecutive calls).
    Entries with equal timestamps may be ordered differently from numpy's argsort (its tie
    order is implementation defined), so rows are compared as time-sorted multisets."""
    g = _graph(samp, 'uniform')
    for rep in range(2):
        res = g.sample_temporal_neighbor(samp['q_nids'], samp['q_ts'], K)
        exp = [samp[f'uniform_K{K}_rep{rep}_{nm}'] for nm in ('nbr', 'eid', 'ts', 'dir')]
        np.testing.assert_array_equal(res[2], exp[2])  # timestamps: sorted, identical
        for r in range(len(res[0])):
            got = sorted(zip(res[2][r].tolist(), res[1][r].tolist(), res[0][r].tolist(), res[3][r].tolist()))
            want = sorted(zip(exp[2][r].tolist(), exp[1][r].tolist(), exp[0][r].tolist(), exp[3][r].tolist()))
            assert got == want, (rep, r)


def test_history_float32_queries(samp):
    g = _graph(samp, 'recent_edges')
    res = g.get_history(samp['q_nids'], samp['q_ts'].astype(np.float32), 8)
    for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
        np.testing.assert_array_equal(a, samp[f'hist32_H8_{nm}'])


@pytest.mark.parametrize('i', [0, 1, 2])
def test_select_latest(samp, i):
    from www2023tiger_amd import hip_ops
    u, idx = hip_ops.select_latest_nids(torch.from_numpy(samp[f'sel{i}_ids']).to(dev()),
                                        torch.from_numpy(samp[f'sel{i}_ts']).to(dev()))
    np.testing.assert_array_equal(u.cpu().numpy(), samp[f'sel{i}_unique'])
    np.testing.assert_array_equal(idx.cpu().numpy(), samp[f'sel{i}_index'])


def test_select_latest_large_bitmap_multiblock():
    """n_nodes large enough for the three-phase scan (more than one 2048-word tile)."""
    from oracle import tiger_oracle as O
    from www2023tiger_amd import hip_ops
    rs = np.random.RandomState(0)
    n_nodes = 700_000
    ids = rs.randint(0, n_nodes, 50_000).astype(np.int64)
    ids[:5000] = rs.randint(0, 300, 5000)
    ts = np.floor(rs.uniform(0, 50, len(ids)))
    u, idx = hip_ops.select_latest_nids(torch.from_numpy(ids).to(dev()), torch.from_numpy(ts).to(dev()), n_nodes)
    eu, ei = O.select_latest_nids(ids, ts)
    np.testing.assert_array_equal(u.cpu().numpy(), eu)
    np.testing.assert_array_equal(idx.cpu().numpy(), ei)


def test_anonymized_reindex(samp):
    from www2023tiger_amd import hip_ops
    for k in ('anon', 'anon2'):
        out = hip_ops.anonymized_reindex(torch.from_numpy(samp[f'{k}_in']).to(dev()))
        np.testing.assert_array_equal(out.cpu().numpy(), samp[f'{k}_out'])


def test_empty_inputs():
    from www2023tiger_amd import hip_ops
    z = load('sampler')
    g = _graph(z, 'recent_edges')
    res = g.sample_temporal_neighbor(np.zeros(0, dtype=np.int64), np.zeros(0), 5)
    assert all(r.shape == (0, 5) for r in res)
    u, idx = hip_ops.select_latest_nids(torch.zeros(0, dtype=torch.int64, device=dev()),
                                        torch.zeros(0, device=dev()))
    assert len(u) == 0 and len(idx) == 0


# ------------------------------------------------------------------------------ dense kernels
@pytest.mark.parametrize('n,in_f,out_f', [(1, 8, 4), (37, 172, 344), (300, 516, 172), (129, 860, 516), (64, 64, 64),
                                          (140001, 44, 300),    # 1 094 x 5 blocks of 128 x 64: the register-blocked form, ragged edges
                                          # activation-stationary blocks (short K of 4 / 6 / 8 k-tiles, >= 8 column tiles), ragged edges
                                          (500, 172, 1032), (70, 100, 600), (129, 256, 520),
                                          # LDS-free short-K blocks (k_gemm_direct: K <= 192, chunk slots 7 / 11 / 12; wave
                                          # tiles 32 x 32 and 32 x 48; rows / columns / K not multiples of the tile sizes)
                                          (1060, 172, 1032), (3, 4, 5), (33, 20, 17), (1815, 100, 400), (257, 192, 70), (40, 180, 1000),
                                          # LDS-free K-split blocks (k_gemm_ks16: K >= 512, 16 / 32 / 48-row blocks), ragged everything
                                          (3072, 1204, 172), (600, 1204, 172), (1500, 516, 172), (47, 1028, 50), (100, 700, 9),
                                          (6144, 500, 100), (3071, 400, 64), (601, 500, 100), (700, 388, 112), (333, 404, 33),
                                          # panel-stationary blocks of eight wavefronts (k_gemm_astat8: 4 / 6 / 8 k-tiles, >= 4 096
                                          # tiles of 128 x 64), ragged rows and columns
                                          (70001, 172, 1032), (140001, 100, 300), (33000, 256, 1024),
                                          # K-split blocks from K = 320 when they fit one round (the score head's product) and
                                          # the 64 x 64 blocks just past it; 128 x 128 register-blocked blocks (N % 128 == 0, K >= 512)
                                          (2048, 344, 172), (400, 364, 172), (3073, 344, 172), (140001, 516, 256), (131073, 640, 128),
                                          # weight-stationary LDS-free blocks (k_gemm_wstat: K <= 256, N <= 256, >= 4 096 tiles of
                                          # 128 x 64): chunk rings of 8 / 11 / 8-of-16 registers, ragged rows, ragged column group
                                          (196613, 256, 256), (262149, 128, 172), (300001, 172, 100), (140001, 200, 250)])
def test_linear_fwd_vs_torch(n, in_f, out_f):
    from www2023tiger_amd.model.dense import linear_forward
    torch.manual_seed(n)
    layer = torch.nn.Linear(in_f, out_f)
    x = torch.randn(n, in_f)
    ref = torch.relu(layer(x)).detach().numpy()
    out = linear_forward(layer.to(dev()), x.to(dev()), relu=True).cpu().numpy()
    assert rel_err(out, ref) < 1e-5


def test_operator_path_modules_run_on_the_library_under_no_grad():
    """MergeLayer / Linear- / MLPMessageFunction forwards (basic_modules.py:17-19, message_modules.py:36,50): under
    no_grad on the GPU they run tg_linear_fwd (one backend for the operator path), with autograd or active dropout
    plain torch; both agree."""
    from www2023tiger_amd.model.basic_modules import MergeLayer
    from www2023tiger_amd.model.message_modules import LinearMessageFunction, MLPMessageFunction
    torch.manual_seed(3)
    ml = MergeLayer(36, 36, 16, 1, dropout=0.1).to(dev()).eval()
    lf, mf = LinearMessageFunction(64, dropout=0.1).to(dev()).eval(), MLPMessageFunction(64, dropout=0.1).to(dev()).eval()
    x1, x2, r = torch.randn(301, 36, device=dev()), torch.randn(301, 36, device=dev()), torch.randn(77, 64, device=dev())
    ref = (ml(x1, x2), lf(r), mf(r))  # grad mode: the autograd functions over the same kernels (unfused ReLU)
    assert all(t.requires_grad for t in ref)
    with torch.no_grad():
        got = (ml(x1, x2), lf(r), mf(r))
    for a, b in zip(got, ref):
        assert a.shape == b.shape and rel_err(a.cpu().numpy(), b.detach().cpu().numpy()) < 1e-5


def test_operator_path_modules_backward_runs_on_the_library():
    """The same modules under autograd (someone training through the per-operator API): forward tg_linear_fwd, backward
    tg_linear_bwd through an autograd function - outputs and every gradient (inputs, weights, biases; the score head's
    d -> 1 layer on zero-padded weights) against plain torch modules carrying the same parameters."""
    import copy
    from www2023tiger_amd.model.basic_modules import MergeLayer
    from www2023tiger_amd.model.message_modules import LinearMessageFunction, MLPMessageFunction
    torch.manual_seed(5)
    for mod, shapes in ((MergeLayer(36, 36, 16, 1, dropout=0.0), ((301, 36), (301, 36))),
                        (MergeLayer(172, 172, 172, 172, dropout=0.0), ((1061, 172), (1061, 172))),
                        (LinearMessageFunction(64, dropout=0.0), ((77, 64),)), (MLPMessageFunction(688, dropout=0.0), ((530, 688),))):
        ref = copy.deepcopy(mod)  # stays on the CPU: plain torch
        mod = mod.to(dev())
        xs_ref = [torch.randn(*sh, requires_grad=True) for sh in shapes]
        xs = [x.detach().to(dev()).requires_grad_(True) for x in xs_ref]
        y_ref, y = ref(*xs_ref), mod(*xs)
        g = torch.randn_like(y_ref)
        y_ref.backward(g)
        y.backward(g.to(dev()))
        assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-5
        for a, b in zip(xs, xs_ref):
            assert rel_err(a.grad.cpu().numpy(), b.grad.numpy()) < 1e-5
        for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
            assert rel_err(p.grad.cpu().numpy(), q.grad.numpy()) < 2e-5, k
    from www2023tiger_amd.model import dense
    assert dense.hip_autograd(xs[0], mod.fn[1])  # (the library's path was the one taken)


@pytest.mark.parametrize('n,d,xw', [
    (1, 8, 32), (200, 172, 688), (131, 100, 400), (513, 16, 60),
    # block configurations of the fused cell: 96-row blocks (+ 16-column tail blocks) when everything fits one
    # round, 128-row blocks otherwise; hidden widths with no partial tile (32, 160), a tail of 4 / 12 / 16 columns
    # (100, 44, 172, 176) and one above 16 (180: padded 32-column tile); row counts at tile and XCD-dealing edges
    (97, 32, 64), (96, 44, 128), (1000, 160, 640), (1153, 176, 256), (600, 180, 360), (4174, 172, 688),
    (4400, 172, 688), (2305, 100, 400), (145, 12, 48),
    # 16-column blocks (k_gru_direct16): 16 / 32 / 48-row blocks (first instance) and 64 / 96-row blocks (second), row counts
    # at the switch points of d = 172 (368 / 736 / 1104 / 1472 rows), one past the largest single round (two tiles per block)
    (368, 172, 688), (369, 172, 688), (737, 172, 688), (1060, 172, 688), (1105, 172, 688), (1473, 172, 688), (2209, 172, 688),
    (1815, 100, 300), (50, 256, 768),
    # 128-row blocks of four wavefronts, two per CU (k_gru<4, 1>: more blocks than CUs; the old-memory tile of the epilogue
    # is captured in registers) - ragged rows, padded / tail column tiles
    (6001, 172, 688), (9000, 256, 1024), (30001, 44, 128), (20000, 100, 300)])
def test_gru_fwd_vs_torch(n, d, xw):
    from www2023tiger_amd.model.dense import gru_forward
    torch.manual_seed(d)
    cell = torch.nn.GRUCell(xw, d)
    x, h = torch.randn(n, xw), torch.randn(n, d)
    ref = cell(x, h).detach().numpy()
    out = gru_forward(cell.to(dev()), x.to(dev()), h.to(dev())).cpu().numpy()
    assert rel_err(out, ref) < 1e-5


def test_time_encode_rounding():
    """cos(fl32(dt*w)+phi) at large dt: an FMA-contracted product is off by up to 4e-2."""
    from www2023tiger_amd import hip_ops
    d = 172
    w = torch.from_numpy((1 / 10 ** np.linspace(0, 9, d)).astype(np.float32))
    phi = torch.linspace(-0.5, 0.5, d)
    ts = torch.tensor([0.0, 1.0, 12345.678, 2.3e6, 2.68e6, 1.37e8])
    ref = torch.cos(ts.unsqueeze(-1) * w + phi).numpy()
    out = hip_ops.time_encode(ts.to(dev()), w.to(dev()), phi.to(dev())).cpu().numpy()
    assert np.abs(out - ref).max() < 5e-7


# ------------------------------------------------------------------------------ collation
@pytest.mark.parametrize('name', MODEL_FIXTURES + TWO_LAYER_FIXTURES)
def test_collator_bit_exact(name):
    z = load(name)
    cfg = parse_cfg(z)
    _, g, coll = build_hip_model(z, cfg)
    B, L = cfg['B'], cfg.get('L', 1)
    for b in range(n_batches(z)):
        sl = slice(b * B, min((b + 1) * B, len(z['src'])))
        out = coll.collate_arrays(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
        cg = out[-1]
        tag = f'b{b}'
        assert len(cg.layers) == L + 1
        np.testing.assert_array_equal(cg.layers[L][0].cpu().numpy(), z[f'{tag}_l1_nids'])
        np.testing.assert_array_equal(cg.layers[L][1].cpu().numpy(), z[f'{tag}_l1_eids'])
        np.testing.assert_array_equal(cg.layers[L][2].cpu().numpy(), z[f'{tag}_l1_ts'])
        if L == 2:  # the deepest hop, sampled at the neighbours' float32 timestamps (data_loader.py:131)
            for j, nm in enumerate(('nids', 'eids', 'ts')):
                np.testing.assert_array_equal(cg.layers[1][j].cpu().numpy(), z[f'{tag}_hop2_{nm}'], err_msg=nm)
        np.testing.assert_array_equal(cg.np_computation_graph_nodes, z[f'{tag}_involved'])
        li = cg.local_index.cpu().numpy()
        np.testing.assert_array_equal(li[z[f'{tag}_involved']], np.arange(len(z[f'{tag}_involved'])))
        rd = cg.restart_data
        np.testing.assert_array_equal(rd.index.cpu().numpy(), z[f'{tag}_rd_index'])
        np.testing.assert_array_equal(rd.nids.cpu().numpy(), z[f'{tag}_rd_nids'])
        np.testing.assert_array_equal(rd.ts.cpu().numpy(), z[f'{tag}_rd_ts'])
        if cfg['restarter'] == 'seq':
            for f, k in (('hist_nids', 'rd_hist_nids'), ('anonymized_ids', 'rd_anon'), ('hist_eids', 'rd_hist_eids'),
                         ('hist_ts', 'rd_hist_ts'), ('hist_dirs', 'rd_hist_dirs')):
                np.testing.assert_array_equal(getattr(rd, f).cpu().numpy(), z[f'{tag}_{k}'], err_msg=k)
        else:
            np.testing.assert_array_equal(rd.prev_ts.cpu().numpy(), z[f'{tag}_rd_prev_ts'])
        for f in ('src_hits', 'dst_hits', 'neg_src_hits', 'neg_dst_hits'):
            np.testing.assert_array_equal(getattr(cg.hit_data, f).cpu().numpy(), z[f'{tag}_{f}'], err_msg=f)


# ------------------------------------------------------------------------------ full stream
def check_state(model, z, tag):
    L, R, S = model.left_memory, model.right_memory, model.msg_store
    assert_close(L.vals.cpu().numpy(), z[f'{tag}_left_vals'], 'left memory', TOL)
    assert_close(R.vals.cpu().numpy(), z[f'{tag}_right_vals'], 'right memory', TOL)
    np.testing.assert_array_equal(L.update_ts.cpu().numpy(), z[f'{tag}_left_ts'])
    np.testing.assert_array_equal(R.update_ts.cpu().numpy(), z[f'{tag}_right_ts'])
    has = np.array(sorted(S.nodes_with_messages), dtype=np.int64)
    np.testing.assert_array_equal(has, z[f'{tag}_has_msg'])
    assert_close(S.node_msg_vals.cpu().numpy()[has], z[f'{tag}_msg_vals'], 'mailbox', TOL)
    np.testing.assert_array_equal(S.node_msg_ts.cpu().numpy()[has], z[f'{tag}_msg_ts'])


def run_stream(name, fused, op_path=False, eager=False):
    z = load(name)
    cfg = parse_cfg(z)
    model, g, coll = build_hip_model(z, cfg)
    if eager:  # updater rows precomputed when a message is stored; restart / flush below force rebuilds of the table
        model.eager_updates()
    if op_path:  # operator-by-operator evaluation (one C call per reference method) instead of the one-call step
        model._fused_eval_ok = lambda *a: False
    B = cfg['B']
    restarting, uptodate = False, set()
    for b in range(n_batches(z)):
        sl = slice(b * B, min((b + 1) * B, len(z['src'])))
        src, dst, neg, ts, eids = (z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids'))
        tag = f'b{b}'
        s_t, d_t, n_t, t_t, e_t, _, cg = coll.collate_arrays(src, dst, neg, ts, eids)
        if b == cfg.get('restart_at', -1):  # lazy restart, train_self_supervised.py:152-163
            restarting, uptodate = True, set()
            model.msg_store.clear()
        if restarting:
            r = np.array(sorted(set(cg.np_computation_graph_nodes.tolist()) - uptodate), dtype=np.int64)
            np.testing.assert_array_equal(r, z[f'{tag}_restart_nids'])
            r_nids = torch.from_numpy(r).to(dev())
            r_ts = torch.full((len(r),), float(np.float32(ts.min())), device=dev())
            if len(r):
                hl, hr, pt = model.restarter_fn(r_nids, r_ts)
                assert_close(hl.cpu().numpy(), z[f'{tag}_restart_h_left'], 'restart h_left', TOL)
                assert_close(hr.cpu().numpy(), z[f'{tag}_restart_h_right'], 'restart h_right', TOL)
                np.testing.assert_array_equal(pt.cpu().numpy(), z[f'{tag}_restart_prev_ts'])
            model.restart(r_nids, r_ts)
            uptodate.update(r.tolist())
            check_state(model, z, f'{tag}_afterrestart')
        if fused:
            buf = model.stream_step(src, dst, neg, ts, eids, want_prev=True)
            nb = len(src)
            counts = buf.counts.cpu().numpy()
            np.testing.assert_array_equal(buf.l1_nids.cpu().numpy(), z[f'{tag}_l1_nids'])
            np.testing.assert_array_equal(buf.l1_eids.cpu().numpy(), z[f'{tag}_l1_eids'])
            np.testing.assert_array_equal(buf.l1_ts.cpu().numpy(), z[f'{tag}_l1_ts'])
            np.testing.assert_array_equal(buf.involved.cpu().numpy()[:counts[0]], z[f'{tag}_involved'])
            assert counts[2] == len(z[f'{tag}_rd_nids'])
            assert_close(buf.h[:2 * nb].cpu().numpy(), z[f'{tag}_h_left'], 'h_left', TOL)
            assert_close(buf.h_prev_left.cpu().numpy(), z[f'{tag}_h_prev_left'], 'h_prev_left', TOL)
            assert_close(buf.h_prev_right.cpu().numpy(), z[f'{tag}_h_prev_right'], 'h_prev_right', TOL)
        else:
            loss, h_left, ps, ns, hpl, hpr = model.contrast_learning(s_t, d_t, n_t, t_t, e_t, cg)
            for k, v in (('h_left', h_left), ('h_prev_left', hpl), ('h_prev_right', hpr)):
                assert_close(v.cpu().numpy(), z[f'{tag}_{k}'], k, TOL)
            for k, v in (('pos_scores', ps), ('neg_scores', ns)):   # one logit per event: the tensor measure
                assert rel_err(v.cpu().numpy(), z[f'{tag}_{k}']) < TOL, (b, k)
            assert abs(float(loss) - float(z[f'{tag}_loss'])) < 1e-4
            # mutual-learning surrogate on the collated restart data (tiger.py:576-590)
            idx = cg.restart_data.index
            u_n = torch.cat([s_t, d_t]).to(dev())[idx]
            u_t = t_t.to(dev()).repeat(2)[idx]
            sl_, sr_, spt = model.restarter_fn(u_n, u_t, cg)
            assert_close(sl_.cpu().numpy(), z[f'{tag}_sur_left'], 'surrogate left', TOL)
            assert_close(sr_.cpu().numpy(), z[f'{tag}_sur_right'], 'surrogate right', TOL)
            np.testing.assert_array_equal(spt.cpu().numpy().reshape(z[f'{tag}_sur_prev_ts'].shape),
                                          z[f'{tag}_sur_prev_ts'])
        if f'{tag}_left_vals' in z.files:
            check_state(model, z, tag)
    model.flush_msg()
    check_state(model, z, 'flushed')


@pytest.mark.parametrize('name', MODEL_FIXTURES + TWO_LAYER_FIXTURES)
def test_stream_reference_api(name):
    """GraphCollator -> TIGER.contrast_learning / restart / flush_msg, the reference's call sequence
    (evaluation takes the one-call step where the configuration allows it)."""
    run_stream(name, fused=False)


@pytest.mark.parametrize('name', MODEL_FIXTURES + TWO_LAYER_FIXTURES)
def test_stream_reference_api_operator_path(name):
    """the same sequence with every reference method bound to its own C entry point"""
    run_stream(name, fused=False, op_path=True)


@pytest.mark.parametrize('eager', [False, True], ids=['lazy', 'eager'])
@pytest.mark.parametrize('name', MODEL_FIXTURES)
def test_stream_fused_step(name, eager):
    """tg_stream_step (collate + STEP 1-6 behind one C call), the benchmarked path; `eager`: with the updater
    run once per stored message (TIGE.eager_updates) instead of on the fly for every involved node."""
    run_stream(name, fused=True, eager=eager)


@pytest.mark.parametrize('eager', [False, True], ids=['lazy', 'eager'])
@pytest.mark.parametrize('name', [n for n in MODEL_FIXTURES if 'restart_at' in parse_cfg(load(n))])
def test_stream_fused_step_with_the_lazy_restart_loop_on_the_device(name, eager):
    """train_self_supervised.py:152-163 without the host's sets: StepBuffers.enable_lazy_restart with the fixture's
    restart batch as the one trigger.  Static restarter: the loop body runs inside tg_stream_step.  Sequence restarter
    (the reference's default recipe, init_utils.py:55-58): the bookkeeping runs on the device (list form), one count is
    read back, SeqRestarter + tg_restart_apply run on the device-resident list.  Against the reference's own loop
    (fixtures): restarted node sets by size, embeddings and state after every batch."""
    z = load(name)
    cfg = parse_cfg(z)
    model, g, coll = build_hip_model(z, cfg)
    if eager:
        model.eager_updates()
    B, nb = cfg['B'], n_batches(z)
    trigger = np.zeros(nb, dtype=np.uint8)
    trigger[cfg['restart_at']] = 1
    buf = model.step_buffers(B, True).enable_lazy_restart(model, trigger)
    seq = cfg['restarter'] == 'seq'
    restarting = False
    for b in range(nb):
        sl = slice(b * B, min((b + 1) * B, len(z['src'])))
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        if len(a[0]) != B:
            break  # (the loop's buffers are of one batch size)
        tag = f'b{b}'
        restarting = restarting or b == cfg['restart_at']
        got = model.stream_step(*a, want_prev=True)
        assert got is buf
        n_ref = len(z[f'{tag}_restart_nids']) if restarting else 0
        n_got = buf.lazy_restarted if seq else int(buf.counts[3].item())
        assert n_got == n_ref, (b, n_got, n_ref)
        assert_close(buf.h[:2 * B].cpu().numpy(), z[f'{tag}_h_left'], 'h_left', TOL)
        assert_close(buf.h_prev_left.cpu().numpy(), z[f'{tag}_h_prev_left'], 'h_prev_left', TOL)
        if f'{tag}_left_vals' in z.files:
            check_state(model, z, tag)


def test_two_layer_model_trains_through_the_drop_in_api():
    """--n_layers 2 (the constructor default, tiger.py:29) in training mode on the reference call sequence: the loss of
    contrast_learning carries the autograd node that hands the device step's gradients over - for BOTH attention layers
    (temporal_embedding_fn.fns.0 / fns.1); the first batch's loss and gradients of the reference's own two-layer run."""
    z = load('train_static_lr_d8_L2')
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    a = [z[k][:cfg['B']] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
    model.train()
    s_t, d_t, n_t, t_t, e_t, _, cg = coll.collate_arrays(*a)
    loss, *_ = model.contrast_learning(s_t, d_t, n_t, t_t, e_t, cg)
    loss.backward()
    assert abs(float(loss) - float(z['b0_contrast_loss'])) < 1e-4
    named = dict(model.named_parameters())
    for k in ('temporal_embedding_fn.fns.0.merger.fc1.weight', 'temporal_embedding_fn.fns.1.merger.fc1.weight',
              'temporal_embedding_fn.fns.1.mha_fn.k_proj_weight'):
        assert named[k].grad is not None and float(named[k].grad.abs().max()) > 0, k


@pytest.mark.parametrize('form', ['lazy', 'eager', 'eager-fused', 'eager-fused-lean'])
@pytest.mark.parametrize('name', TWO_LAYER_FIXTURES)
def test_stream_fused_step_two_layers(name, form):
    """--n_layers 2 inside tg_stream_step (tg_step_io.inner): the second hop sampled in the step at the neighbours'
    float32 timestamps (data_loader.py:128-131), the Q*K neighbour slots embedded with fns[1] at the root's query time
    and fed to fns[0] as the node part of its keys (temporal_agg_modules.py:57-66) - against the reference's own two-layer
    streams, restart and flush included; with the lazy / eager updater, pre-multiplied weights for both layers, and the
    lean form (no involved set formed)."""
    z = load(name)
    cfg = parse_cfg(z)
    model, g, coll = build_hip_model(z, cfg)
    if 'eager' in form:
        model.eager_updates()
    if 'fused' in form:
        model.fuse_attention()
        assert model.model_struct(0).attn_fused and model.model_struct(1).attn_fused
    lean = 'lean' in form
    B = cfg['B']
    restarting, uptodate = False, set()
    for b in range(n_batches(z)):
        sl = slice(b * B, min((b + 1) * B, len(z['src'])))
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        tag = f'b{b}'
        if b == cfg.get('restart_at', -1):
            restarting, uptodate = True, set()
            model.msg_store.clear()
        if restarting:
            r = np.array(sorted(set(z[f'{tag}_involved'].tolist()) - uptodate), dtype=np.int64)
            np.testing.assert_array_equal(r, z[f'{tag}_restart_nids'])
            model.restart(torch.from_numpy(r).to(dev()), torch.full((len(r),), float(np.float32(a[3].min())), device=dev()))
            uptodate.update(r.tolist())
        buf = model.stream_step(*a, lean=lean)
        nb = len(a[0])
        counts = buf.counts.cpu().numpy()
        np.testing.assert_array_equal(buf.l1_nids.cpu().numpy(), z[f'{tag}_l1_nids'])
        if not lean:
            np.testing.assert_array_equal(buf.involved.cpu().numpy()[:counts[0]], z[f'{tag}_involved'])  # hop 2 included
        assert_close(buf.h[:2 * nb].cpu().numpy(), z[f'{tag}_h_left'], 'h_left', TOL)
        if f'{tag}_left_vals' in z.files:
            check_state(model, z, tag)
    model.flush_msg()
    check_state(model, z, 'flushed')


@pytest.mark.parametrize('strategy', ['recent_nodes', 'uniform'])
@pytest.mark.parametrize('form', ['lazy', 'eager-fused-lean'])
def test_stream_fused_step_two_layers_recent_nodes_strategy(form, strategy):
    """--n_layers 2 with --strategy recent_nodes / uniform inside tg_stream_step: BOTH hops follow the graph's strategy
    (data_loader.py:128-131 samples every layer with graph.sample_temporal_neighbor), the second at the neighbours' float32
    timestamps - uniform: the graph's MT19937 stream goes on behind the first hop's draws - lists of both hops bit-exact,
    embeddings and state against the oracle collating the same way."""
    from oracle import tiger_oracle as O
    from test_oracle_golden import build_oracle
    z = load('static_lr_d8_L2')
    cfg = parse_cfg(z)
    model, g, coll = build_hip_model(z, cfg, strategy=strategy)
    orc = build_oracle(z, cfg)
    orc.graph = O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'], strategy=strategy, seed=0)
    if 'eager' in form:
        model.eager_updates()
        model.fuse_attention()
    B, differs = cfg['B'], False
    for b in range(n_batches(z)):
        sl = slice(b * B, min((b + 1) * B, len(z['src'])))
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], cfg['K'], 'static', n_layers=2)
        ref = orc.contrast_learning(*a, cg)['h_left'].detach().numpy()
        buf = model.stream_step(*a, lean='lean' in form)
        np.testing.assert_array_equal(buf.l1_nids.cpu().numpy(), cg['l1_nids'])
        np.testing.assert_array_equal(buf.l1_eids.cpu().numpy(), cg['l1_eids'])
        if 'lean' not in form:
            cnt = buf.counts.cpu().numpy()
            np.testing.assert_array_equal(buf.involved.cpu().numpy()[:cnt[0]], cg['involved'])  # second hop included
        assert_close(buf.h[:2 * len(a[0])].cpu().numpy(), ref, 'h_left', TOL)
        edges = orc.graph.sample_temporal_neighbor(cg['l1_nids'].ravel(), cg['l1_ts'].ravel(), cfg['K'], strategy='recent_edges')[0]
        differs = differs or not np.array_equal(edges, cg['hop2_nids'])
    assert differs  # the strategy really changes the second hop on this stream
    compare_state_with_oracle(model, orc)


@pytest.mark.parametrize('name,strategy', [('static_lr_d8_L2', 'recent_edges'), ('static_lr_d8_L2', 'recent_nodes'),
                                           ('seq_lr_d8', 'recent_nodes')])
def test_collate_only_pass_lists_what_the_step_involves(name, strategy):
    """The list form of the lazy restart (tg_lazy_restart.list: a collate-only pass lists `involved & ~uptodate`): with nobody
    up to date the list IS the batch's involved set - the nodes of EVERY layer, sampled with the graph's own strategy
    (data_loader.py:105-131) - against the oracle's collation; and not the set another strategy / one hop would give."""
    import ctypes as C
    from oracle import tiger_oracle as O
    from www2023tiger_amd._lib import check, lib, ptr
    from www2023tiger_amd.hip_ops import stream_ptr
    z = load(name)
    cfg = parse_cfg(z)
    L = cfg.get('L', 1)
    model, g, _ = build_hip_model(z, cfg, strategy=strategy)
    og = O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'], strategy=strategy, seed=0)
    other = O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'], strategy='recent_edges', seed=0)
    B, differs = cfg['B'], False
    for b in range(2, n_batches(z)):
        sl = slice(b * B, min((b + 1) * B, len(z['src'])))
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        buf = model.step_buffers(len(a[0]))
        buf.enable_lazy_restart(model, np.zeros(1, dtype=np.uint8), force_list=True)
        buf.lazy_restarting.fill_(1)
        to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
        buf.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64), to(a[4], torch.int64))
        cb = buf._lazy_collate
        model.prepare_pass(cb, g)
        m = model.model_struct()
        check(lib.tg_stream_step(C.byref(m), C.byref(g.tcsr), C.byref(cb.io), ptr(cb.ws), cb.ws.numel(), stream_ptr(dev())), 'pass')
        got = sorted(cb.lazy_list[:int(cb.counts[3])].tolist())
        ref = O.collate(og, a[0], a[1], a[2], a[3], cfg['K'], 'static', n_layers=L)['involved']
        assert got == sorted(set(ref.tolist())), b
        one_hop = O.collate(other, a[0], a[1], a[2], a[3], cfg['K'], 'static', n_layers=1)['involved']
        differs = differs or sorted(set(one_hop.tolist())) != got
    assert differs  # (second hop / strategy do change the set on this stream)


def test_no_feat_buffer_reads_pinned_host_tables():
    """--no_feat_buffer (feature_getter.py:41-47,86-87): NumericalFeature(register_buffer=False) keeps the feature tables
    in pinned host memory; the kernels read them in place, the stream reproduces the reference like the resident form"""
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.model.feature_getter import NumericalFeature
    from www2023tiger_amd.model.restarters import StaticRestarter
    from www2023tiger_amd.model.tiger import TIGER
    z = load('static_ll_d16')
    cfg = parse_cfg(z)
    n_nodes, nfeats, efeats = fixture_tables(z, cfg)
    g = Graph.from_arrays(z['src'], z['dst'], z['ts'], z['eids'], strategy='recent_edges', seed=0, device=dev())
    fg = NumericalFeature(torch.from_numpy(nfeats), torch.from_numpy(efeats), dim=cfg['d'], register_buffer=False, device=dev())
    fg.n_nodes, fg.n_edges = n_nodes, len(z['src'])
    model = TIGER(raw_feat_getter=fg, graph=g, restarter=StaticRestarter(raw_feat_getter=fg, graph=g), n_neighbors=cfg['K'],
                  hit_type='bin', n_layers=1, n_head=2, dropout=0.1, msg_src=cfg['msg_src'], upd_src=cfg['upd_src'])
    params = fixture_params(z, cfg)
    with torch.no_grad():
        for k, v in model.named_parameters():
            v.copy_(torch.from_numpy(params[k]))
    model = model.to(dev()).eval()
    assert fg.nfeats.device.type == 'cpu' and fg.nfeats.is_pinned() and fg.efeats.is_pinned()   # .to() left them on the host
    assert 'nfeats' not in dict(fg.named_buffers())
    B = cfg['B']
    for b in range(4):
        sl = slice(b * B, (b + 1) * B)
        buf = model.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
        assert_close(buf.h[:2 * B].cpu().numpy(), z[f'b{b}_h_left'], 'h_left', TOL)
    check_state(model, z, 'b3')
    ids = torch.tensor([1, 5, 7], device=dev())
    np.testing.assert_array_equal(fg.get_node_embeddings(ids).cpu().numpy(), nfeats[[1, 5, 7]])


def test_invariant_errors_surface_as_value_errors():
    """memory.py:45-46: writing a memory row back in time must raise, as in the reference."""
    from www2023tiger_amd.model.memory import Memory
    m = Memory(10, 8).to(dev())
    ids = torch.tensor([1, 2], device=dev())
    m.set(ids, torch.ones(2, 8, device=dev()), torch.tensor([5.0, 5.0], device=dev()))
    with pytest.raises(ValueError, match='past memory'):
        m.set(ids, torch.ones(2, 8, device=dev()), torch.tensor([4.0, 6.0], device=dev()))
    with pytest.raises(ValueError, match='Duplicate'):
        m.set(torch.tensor([3, 3], device=dev()), torch.ones(2, 8, device=dev()), torch.tensor([9.0, 9.0], device=dev()))


def _model_for_checkpoint(z, cfg, device):
    """a model of the fixture's configuration with weights that have nothing to do with the checkpoint"""
    from www2023tiger_amd.data.data_loader import GraphCollator
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.model.feature_getter import NumericalFeature
    from www2023tiger_amd.model.restarters import SeqRestarter, StaticRestarter
    from www2023tiger_amd.model.tiger import TIGER
    g = Graph.from_arrays(z['src'], z['dst'], z['ts'], z['eids'], strategy='recent_edges', seed=0, device=device)
    fg = NumericalFeature(torch.from_numpy(z['nfeats']), torch.from_numpy(z['efeats']), dim=cfg['d'], device=device)
    fg.n_nodes, fg.n_edges = int(z['n_nodes']), len(z['src'])
    torch.manual_seed(99)
    rst = (SeqRestarter(raw_feat_getter=fg, graph=g, hist_len=cfg['H'], n_head=2, dropout=0.1) if cfg['restarter'] == 'seq'
           else StaticRestarter(raw_feat_getter=fg, graph=g))
    model = TIGER(raw_feat_getter=fg, graph=g, restarter=rst, n_neighbors=cfg['K'], hit_type=cfg['hit'], n_layers=1,
                  n_head=2, dropout=0.1, msg_src=cfg['msg_src'], upd_src=cfg['upd_src'])
    return model, g, GraphCollator(g, cfg['K'], 1, restarter=cfg['restarter'], hist_len=cfg.get('H'))


@pytest.mark.parametrize('name', ['ckpt_seq_lr_d8', 'ckpt_static_ll_d16'])
def test_reference_checkpoint_loads_and_the_stream_continues(name):
    """train_self_supervised.py:208-209 / 110-114: a state_dict written by the REFERENCE (after flush_msg, every
    key and tensor in the fixture) loads with strict=True - the key set is the reference's, alias keys
    msg_memory.* / upd_memory.* and the shared time encoders included - and the batches that follow reproduce the
    reference's continuation (embeddings, scores, loss, final memories)."""
    z = load(name)
    cfg = parse_cfg(z)
    model, g, coll = _model_for_checkpoint(z, cfg, dev())
    model = model.to(dev()).eval()
    keys = [str(k) for k in z['sd_keys']]
    assert list(model.state_dict().keys()) == keys           # same keys, same order
    sd = {k: torch.from_numpy(z['sd.' + k]) for k in keys}
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    B, at = cfg['B'], cfg['ckpt_at']
    fused = model.__class__.__name__ == 'TIGER'
    for b in range(at, at + cfg['n_after']):
        sl = slice(b * B, (b + 1) * B)
        s_t, d_t, n_t, t_t, e_t, _, cg = coll.collate_arrays(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
        loss, h_left, ps, ns, _, _ = model.contrast_learning(s_t, d_t, n_t, t_t, e_t, cg)
        assert_close(h_left.cpu().numpy(), z[f'b{b}_h_left'], 'h_left', TOL)
        assert rel_err(ps.cpu().numpy(), z[f'b{b}_pos_scores']) < TOL and rel_err(ns.cpu().numpy(), z[f'b{b}_neg_scores']) < TOL
        assert abs(float(loss) - float(z[f'b{b}_loss'])) < 1e-4
    check_state(model, z, 'final')
    # and the streaming path (pre-multiplied weights, eager updates) from the same checkpoint
    m2, _, _ = _model_for_checkpoint(z, cfg, dev())
    m2 = m2.to(dev()).eval()
    m2.fuse_attention()
    m2.eager_updates()
    m2.load_state_dict(sd, strict=True)                       # derived tables follow the new parameters
    for b in range(at, at + cfg['n_after']):
        sl = slice(b * B, (b + 1) * B)
        buf = m2.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
        assert_close(buf.h[:2 * B].cpu().numpy(), z[f'b{b}_h_left'], 'h_left (stream)', TOL)
    check_state(m2, z, 'final')


# ------------------------------------------------------------------------------ larger shapes vs the oracle
def compare_state_with_oracle(model, orc):
    """memories, mailbox rows, has-message set and timestamps of the HIP model against the oracle's"""
    has = model.msg_store.has_msg_mask().cpu().numpy()
    np.testing.assert_array_equal(np.nonzero(has)[0], np.nonzero(orc.has_msg)[0])
    has = np.nonzero(has)[0]
    assert_close(model.left_memory.vals.cpu().numpy(), orc.left_vals.numpy(), 'left memory', TOL)
    assert_close(model.right_memory.vals.cpu().numpy(), orc.right_vals.numpy(), 'right memory', TOL)
    assert_close(model.msg_store.node_msg_vals[torch.from_numpy(has).to(dev())].cpu().numpy(),
                 orc.msg_vals.numpy()[has], 'mailbox', TOL)
    np.testing.assert_array_equal(model.left_memory.update_ts.cpu().numpy(), orc.left_ts.numpy())
    np.testing.assert_array_equal(model.right_memory.update_ts.cpu().numpy(), orc.right_ts.numpy())
    np.testing.assert_array_equal(model.msg_store.node_msg_ts.cpu().numpy()[has], orc.msg_ts.numpy()[has])


def _oracle_vs_fused(stream, d, K, B, n_batches, msg_src, upd_src, zero_nfeats=True, fuse=False, eager=False,
                     lean=None):
    """lean: None = never; 'all' = every step lean; 'mixed' = the SAME step buffers (one workspace) switch between the
    lean and the full form every third batch, which also checks that either form leaves the workspace clean"""
    import bench
    from oracle import tiger_oracle as O
    model, orc = bench.build_models(stream, d, K, msg_src, upd_src, with_oracle=True, zero_nfeats=zero_nfeats)
    if fuse:
        model.fuse_attention()
    if eager:
        model.eager_updates()
    worst = 0.0
    for b in range(n_batches):
        sl = slice(b * B, (b + 1) * B)
        a = [stream[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        is_lean = lean == 'all' or (lean == 'mixed' and b % 3 != 1)
        if lean == 'mixed':
            model.step_buffers(B).io.lean = 1 if is_lean else 0
        buf = model.stream_step(*a, lean=(lean == 'all'))
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        np.testing.assert_array_equal(buf.l1_nids.cpu().numpy(), cg['l1_nids'])
        np.testing.assert_array_equal(buf.l1_eids.cpu().numpy(), cg['l1_eids'])
        counts = buf.counts.cpu().numpy()
        if is_lean:  # the sets were not formed; the number of unique positive nodes still is reported
            assert counts[0] == -1 and counts[1] == -1
            assert counts[2] == len(np.unique(np.concatenate([a[0], a[1]])))
        else:
            np.testing.assert_array_equal(buf.involved.cpu().numpy()[:counts[0]], cg['involved'])
        worst = max(worst, assert_close(buf.h[:2 * B].cpu().numpy(), ref, 'h_left', TOL)[0])
    compare_state_with_oracle(model, orc)
    assert worst < TOL, worst


def test_c2_full_size_batches_match_oracle():
    """BASELINE configs[1] at full size (Wikipedia-shaped, d=172, B=1024, K=10): 5 batches."""
    import bench
    c = bench.C2
    stream = bench.make_stream(c['n_u'], c['n_i'], 20000, c['T'] * 20000 / c['E'], seed=1, d_e=c['d'])
    _oracle_vs_fused(stream, c['d'], c['K'], c['B'], 5, c['msg_src'], c['upd_src'])


def test_c2_bench_configuration_soak_matches_oracle():
    """The benchmarked configuration exactly (C2 shapes, pre-multiplied attention weights, device-built
    graph) over 16 consecutive batches: errors must not accumulate through the memories."""
    import bench
    c = bench.C2
    stream = bench.make_stream(c['n_u'], c['n_i'], 40000, c['T'] * 40000 / c['E'], seed=4, d_e=c['d'])
    _oracle_vs_fused(stream, c['d'], c['K'], c['B'], 16, c['msg_src'], c['upd_src'], fuse=True, eager=True)


@pytest.mark.parametrize('lean', ['all', 'mixed'])
def test_c2_lean_steps_match_oracle(lean):
    """tg_step_io.lean (no involved / outdated sets formed: the compaction launch is skipped, dedup slots are indexed by
    node id, the time invariants are checked per centre and per neighbour) against the oracle, C2 shapes, 12 batches"""
    import bench
    c = bench.C2
    stream = bench.make_stream(c['n_u'], c['n_i'], 40000, c['T'] * 40000 / c['E'], seed=5, d_e=c['d'])
    _oracle_vs_fused(stream, c['d'], c['K'], c['B'], 12, c['msg_src'], c['upd_src'], fuse=True, eager=True, lean=lean)


def test_lean_steps_with_more_node_ids_than_batch_slots():
    """B = 200 on the 9 228-node graph: 3B(K+1) = 6 600 < n_nodes, so the id-indexed dedup slots of a lean step are a table
    of their own (one per node id) and the write-back cleans the positions of the batch's positive nodes, not a prefix"""
    import bench
    c = bench.C2
    stream = bench.make_stream(c['n_u'], c['n_i'], 8000, c['T'] * 8000 / c['E'], seed=7, d_e=c['d'])
    _oracle_vs_fused(stream, c['d'], c['K'], 200, 24, c['msg_src'], c['upd_src'], fuse=True, eager=True, lean='mixed')


def test_lean_step_checks_the_time_invariants():
    """a message older than its node's memory (message_modules.py:158-159) must raise from a lean step as from a full
    one: the check moved from the outdated list to the centres / neighbours of the batch"""
    import bench
    c = bench.C2
    stream = bench.make_stream(c['n_u'], c['n_i'], 8000, c['T'] * 8000 / c['E'], seed=6, d_e=c['d'])
    B = c['B']
    for where in ('centre', 'neighbour'):
        model, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'])
        model.fuse_attention()
        model.eager_updates()
        for b in range(3):
            a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
            model.stream_step(*a, lean=True)
        a = [stream[k][3 * B:4 * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        has = model.msg_store.has_msg_mask().cpu().numpy()
        if where == 'centre':
            v = next(int(x) for x in a[0] if has[x])
        else:  # a node that is only a sampled neighbour in this batch
            g = model.graph
            nb = g.sample_temporal_neighbor(np.asarray(a[0]), np.asarray(a[3]), c['K'])[0]
            centres = set(np.concatenate([a[0], a[1], a[2]]).tolist())
            v = next(int(x) for x in nb.reshape(-1) if x != 0 and has[x] and int(x) not in centres)
        mem = model.left_memory if c['msg_src'] == 'left' else model.right_memory
        mem.update_ts[v] += 1.0e6   # state written behind the model's back: the memory is now ahead of the stored message
        with pytest.raises(ValueError):
            model.stream_step(*a, lean=True)


def test_c2_lazy_and_eager_updates_agree():
    """the same stream through the lazy form (updater on every involved node with a pending message, every
    batch) and the eager form (once per stored message): embeddings and state agree to float32 rounding of
    differently tiled products, far inside the parity bar"""
    import bench
    c = bench.C2
    stream = bench.make_stream(c['n_u'], c['n_i'], 12 * c['B'], c['T'] * 12 * c['B'] / c['E'], seed=8, d_e=c['d'])
    lazy, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'])
    eager, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'])
    eager.eager_updates()
    for b in range(10):
        sl = slice(b * c['B'], (b + 1) * c['B'])
        a = [stream[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        h0, h1 = lazy.stream_step(*a), eager.stream_step(*a)
        np.testing.assert_array_equal(h0.counts.cpu().numpy()[:3], h1.counts.cpu().numpy()[:3])
        assert_close(h1.h.cpu().numpy(), h0.h.cpu().numpy(), 'h (eager vs lazy)', 1e-5)
        if b == 5:   # a flush in the middle: the eager table must be rebuilt (here: emptied) behind it
            lazy.flush_msg(); eager.flush_msg()
    assert_close(eager.left_memory.vals.cpu().numpy(), lazy.left_memory.vals.cpu().numpy(), 'left memory', 1e-5)
    assert_close(eager.right_memory.vals.cpu().numpy(), lazy.right_memory.vals.cpu().numpy(), 'right memory', 1e-5)
    assert torch.equal(eager.msg_store.has_msg_bits, lazy.msg_store.has_msg_bits)


@pytest.mark.parametrize('lean', [False, True], ids=['full', 'lean'])
def test_fused_step_with_the_recent_nodes_strategy_matches_oracle(lean):
    """--strategy recent_nodes (graph.py:129-143: the last occurrence of each distinct neighbour) inside tg_stream_step
    (tg_step_io.strategy = 1): neighbour lists bit-exact, embeddings and state against the oracle collating with the same
    strategy; `uniform` is refused by the step (it runs on the operator path)."""
    import bench
    from oracle import tiger_oracle as O
    from www2023tiger_amd.data.graph import Graph
    c = bench.C2
    B, K, d, nb = 512, c['K'], 64, 8
    stream = bench.make_stream(600, 80, (nb + 1) * B, 3.0e4, seed=31, d_e=d)   # few nodes: neighbours repeat a lot
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], with_oracle=True)
    model.graph = Graph.from_arrays(stream['src'], stream['dst'], stream['ts'], stream['eids'], strategy='recent_nodes',
                                    seed=0, max_node_id=stream['n_nodes'] - 1, device=dev())
    orc.graph = O.OracleGraph(stream['src'], stream['dst'], stream['ts'], stream['eids'], strategy='recent_nodes',
                              max_node_id=stream['n_nodes'] - 1)
    model.fuse_attention()
    model.eager_updates()
    differs = False
    for b in range(nb):
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        buf = model.stream_step(*a, lean=lean)
        np.testing.assert_array_equal(buf.l1_nids.cpu().numpy(), cg['l1_nids'])
        np.testing.assert_array_equal(buf.l1_eids.cpu().numpy(), cg['l1_eids'])
        np.testing.assert_array_equal(buf.l1_ts.cpu().numpy(), cg['l1_ts'])
        if not lean:
            cnt = buf.counts.cpu().numpy()
            np.testing.assert_array_equal(buf.involved.cpu().numpy()[:cnt[0]], cg['involved'])
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, 'h_left', TOL)
        edges = orc.graph.sample_temporal_neighbor(np.concatenate(a[:3]), np.tile(a[3], 3), K, strategy='recent_edges')[0]
        differs = differs or not np.array_equal(edges, cg['l1_nids'])
    assert differs  # the two strategies really sample different lists on this stream
    compare_state_with_oracle(model, orc)


@pytest.mark.parametrize('lean', [False, True], ids=['full', 'lean'])
def test_fused_step_with_the_uniform_strategy_matches_oracle(lean):
    """--strategy uniform (graph.py:101-115) inside tg_stream_step (tg_step_io.strategy = 2): K draws of numpy's legacy
    randint per non-empty query, consumed from the graph's MT19937 stream in query order (cat[src, dst, neg]) and sorted by
    time.  The oracle's graph carries a numpy RandomState with the same seed: every batch's neighbour lists must be the very
    same draws (node ids / edge ids / times equal as time-sorted multisets - numpy's argsort leaves the order of equal times
    open), the generator must stand where numpy's stands after the stream, and embeddings and state follow."""
    import bench
    from oracle import tiger_oracle as O
    from www2023tiger_amd.data.graph import Graph
    c = bench.C2
    B, K, d, nb = 256, c['K'], 32, 6
    stream = bench.make_stream(400, 60, (nb + 1) * B, 2.0e4, seed=43, d_e=d)
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], with_oracle=True)
    model.graph = Graph.from_arrays(stream['src'], stream['dst'], stream['ts'], stream['eids'], strategy='uniform',
                                    seed=5, max_node_id=stream['n_nodes'] - 1, device=dev())
    orc.graph = O.OracleGraph(stream['src'], stream['dst'], stream['ts'], stream['eids'], strategy='uniform', seed=5,
                              max_node_id=stream['n_nodes'] - 1)
    model.fuse_attention()
    model.eager_updates()
    for b in range(nb):
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        buf = model.stream_step(*a, lean=lean)
        got_t, got_n, got_e = (x.cpu().numpy() for x in (buf.l1_ts, buf.l1_nids, buf.l1_eids))
        np.testing.assert_array_equal(got_t, cg['l1_ts'])  # times are sorted: equal as arrays
        np.testing.assert_array_equal(np.sort(got_e, axis=1), np.sort(cg['l1_eids'], axis=1))  # the same draws
        key = lambda n, e: np.sort(n.astype(np.int64) * (int(stream['eids'].max()) + 1) + e, axis=1)
        np.testing.assert_array_equal(key(got_n, got_e), key(cg['l1_nids'], cg['l1_eids']))
        assert_close(buf.h[:2 * B].cpu().numpy(), ref, f'h_left, batch {b}', TOL)
    # the generator stands where numpy's stands: the next draws of both agree
    st = model.graph._mt_state().cpu().numpy().view(np.uint32)
    ref_state = orc.graph.rng.get_state()
    np.testing.assert_array_equal(st[:624], ref_state[1])
    assert int(st[624]) == int(ref_state[2])
    compare_state_with_oracle(model, orc)


@pytest.mark.parametrize('mc', ['12', '16'])
def test_one_launch_attention_tile_kernel_matches_oracle(mc):
    """k_attn_tile (the whole attention block of a tile of centres in one workgroup, G / S in LDS only) is off by default -
    measured slower than the four launches it replaces - and stays parity-tested: TG_ATTN_TILE=1 is read when the library
    first launches an attention block, so the run is a child process.  Tiles of 12 centres (12 wavefronts) and of 16
    (8 wavefronts, two centres each), C2 shapes, eager + lean steps against the oracle."""
    import os
    import subprocess
    import sys
    code = ('import sys; sys.path.insert(0, "tests"); import bench, test_hip_parity as t; '
            'from www2023tiger_amd._lib import lib; import ctypes as C; c = bench.C2; '
            'stream = bench.make_stream(c["n_u"], c["n_i"], 12000, c["T"] * 12000 / c["E"], seed=41, d_e=c["d"]); '
            't._oracle_vs_fused(stream, c["d"], c["K"], c["B"], 6, c["msg_src"], c["upd_src"], fuse=True, eager=True, lean="mixed"); '
            'm, _ = bench.build_models(stream, c["d"], c["K"], c["msg_src"], c["upd_src"]); m.fuse_attention(); '
            'assert lib.tg_attn_tile_applies(C.byref(m.model_struct())) == 1; print("tile parity ok")')
    env = dict(os.environ, TG_ATTN_TILE='1', TG_ATTN_TILE_MC=mc)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'tile parity ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_derived_tables_follow_in_place_writes_to_state_and_parameters():
    """The eager-update table and the pre-multiplied attention weights are functions of state / parameters.  In-place torch
    writes to a memory tensor, to the mailbox, to an updater weight or to an attention weight between steps (no
    Memory.set, no invalidate_pending(), no second fuse_attention()) are seen through the tensors' version counters and
    the derived tables are rebuilt: the eager + fused model keeps agreeing with a lazy, unfused one given the same
    writes."""
    import bench
    c = bench.C2
    B = 256
    stream = bench.make_stream(c['n_u'], c['n_i'], 12 * B, c['T'] * 12 * B / c['E'], seed=33, d_e=c['d'])
    plain, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'])
    fast, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'])
    fast.eager_updates()
    fast.fuse_attention()
    for b in range(10):
        a = [stream[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        for m in (plain, fast):
            if b == 3:   # state written behind the model's back, through plain torch indexing
                v = torch.as_tensor(a[0][:7], device=dev())
                m.left_memory.vals[v] *= 0.5
                m.right_memory.vals[v] += 0.25
            if b == 5:   # an "optimizer step" on the updater and on the attention block
                with torch.no_grad():
                    m.right_mem_updater.cell.weight_hh.mul_(1.05)
                    m.temporal_embedding_fn.fns[0].merger.fc2.weight.add_(0.01)
                    m.time_encoder.phase.add_(0.05)
        h0, h1 = plain.stream_step(*a), fast.stream_step(*a)
        assert_close(h1.h.cpu().numpy(), h0.h.cpu().numpy(), f'h, batch {b}', 2e-5)
    assert_close(fast.left_memory.vals.cpu().numpy(), plain.left_memory.vals.cpu().numpy(), 'left memory', 2e-5)
    assert_close(fast.right_memory.vals.cpu().numpy(), plain.right_memory.vals.cpu().numpy(), 'right memory', 2e-5)


@pytest.mark.parametrize('B', [1200, 683, 200])
def test_stream_k_piece_sums_with_uneven_row_tiles(B):
    """The merged fc1 product runs as stream-K pieces whenever it has 128..255 tiles (C2: 144).  B = 1200 gives
    57 row tiles (not a multiple of the 8 XCDs the units are dealt over, last tile partly empty), B = 683 gives
    33 row tiles x 3 = 99 tiles and B = 200 (the reference's default batch) 10 x 3 = 30 tiles, each cut into more
    pieces (up to eight per tile); all against the oracle with pre-multiplied weights."""
    import bench
    c = bench.C2
    stream = bench.make_stream(c['n_u'], c['n_i'], 4 * B, c['T'] * 4 * B / c['E'], seed=6, d_e=c['d'])
    _oracle_vs_fused(stream, c['d'], c['K'], B, 3, c['msg_src'], c['upd_src'], fuse=True)


def test_large_sparse_graph_multiblock_compaction_matches_oracle():
    """300 k nodes: the involved / outdated compaction takes the three-phase (multi-block) path;
    no feature tables, d=100 (LastFM-style widths)."""
    import bench
    stream = bench.make_stream(250000, 50000, 12000, 5.0e5, seed=2, d_e=100, with_efeats=False)
    _oracle_vs_fused(stream, 100, 10, 2048, 4, 'left', 'right', zero_nfeats=False)


def test_sampler_properties_at_scale():
    """Size-independent properties on a 1 M-event graph: sampled timestamps are ascending, strictly
    earlier than the query, left padded, and a wider K extends a narrower one on the left."""
    import bench
    from www2023tiger_amd.data.graph import Graph
    st = bench.make_stream(20000, 3000, 1_000_000, 1.0e7, seed=3, with_efeats=False)
    g = Graph.from_arrays(st['src'], st['dst'], st['ts'], st['eids'], strategy='recent_edges', device=dev())
    rs = np.random.RandomState(0)
    q = rs.randint(0, st['n_nodes'], 200_000).astype(np.int64)
    t = rs.uniform(0, 1.05e7, len(q))
    n10, e10, t10, d10 = g.sample_temporal_neighbor(q, t, 10)
    n40, e40, t40, _ = g.sample_temporal_neighbor(q, t, 40)
    np.testing.assert_array_equal(n40[:, -10:], n10)
    np.testing.assert_array_equal(e40[:, -10:], e10)
    valid = n10 != 0
    assert (t10[valid] < t[:, None].repeat(10, 1)[valid].astype(np.float32) + 1e-3).all()
    both = valid[:, 1:] & valid[:, :-1]
    assert (np.diff(t10, axis=1)[both] >= 0).all()                              # ascending in time
    assert ((~valid)[:, 1:] <= (~valid)[:, :-1]).all()                          # pads only on the left
    src_of = {int(e): (int(s), int(dd)) for e, s, dd in zip(st['eids'][:2000], st['src'][:2000], st['dst'][:2000])}
    for i in np.nonzero(valid.any(1))[0][:500]:                                 # edge ids join query node and neighbour
        for k in np.nonzero(valid[i])[0]:
            if int(e10[i, k]) in src_of:
                s, dd = src_of[int(e10[i, k])]
                assert {int(q[i]), int(n10[i, k])} == {s, dd}
                assert d10[i, k] == (1 if int(q[i]) == dd else 0)


@pytest.mark.parametrize('form', ['lazy', 'eager', 'eager-lean'])
def test_ragged_and_degenerate_batches_match_oracle(form):
    """Batch sizes 1, 7 and a ragged tail; a batch whose events all share one timestamp (every
    dedup decision is a tie); the same node as src of many events (collisions).  In the lazy form, with eager
    updates, and as lean eager steps (dedup slots indexed by node id, no involved set)."""
    import bench
    from oracle import tiger_oracle as O
    rs = np.random.RandomState(5)
    st = bench.make_stream(40, 12, 700, 90.0, seed=5, d_e=16)
    st['ts'][300:364] = st['ts'][300]            # 64 simultaneous events
    st['ts'] = np.sort(st['ts'])
    st['src'][400:440] = st['src'][400]          # one hot source node
    model, orc = bench.build_models(st, 16, 5, 'right', 'left', with_oracle=True)
    if form != 'lazy':
        model.eager_updates()
    edges = [0, 1, 8, 72, 300, 364, 400, 440, 571, 700]   # sizes 1, 7, 64, 228, 64, 36, 40, 131, 129
    for lo, hi in zip(edges[:-1], edges[1:]):
        a = [st[k][lo:hi] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        buf = model.stream_step(*a, lean=(form == 'eager-lean'))
        assert (int(buf.counts[0]) == -1) == (form == 'eager-lean')
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], 5, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        n = hi - lo
        np.testing.assert_array_equal(buf.l1_nids.cpu().numpy()[:3 * n], cg['l1_nids'])
        assert_close(buf.h[:2 * n].cpu().numpy(), ref, 'h_left', TOL)
    compare_state_with_oracle(model, orc)


def test_reset_and_memory_snapshots():
    """reset(), save_memory_state / load_memory_state (tiger.py:457-484) rewind the stream exactly."""
    z = load('static_ll_d16')
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg)
    B = cfg['B']
    def run(b):
        sl = slice(b * B, (b + 1) * B)
        return model.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids'))).h.clone()
    run(0); run(1)
    snap = model.save_memory_state()
    h2 = run(2)
    run(3)
    model.load_memory_state(snap)
    h2b = run(2)
    assert torch.equal(h2, h2b)
    model.reset()
    assert float(model.left_memory.vals.abs().sum()) == 0 and not model.msg_store.nodes_with_messages
    h0 = run(0)
    assert rel_err(h0[:2 * B].cpu().numpy(), z['b0_h_left']) < TOL


def test_time_encode_large_arguments():
    """|dt * w + phi| beyond the fast path's 3e6 (float64 range reduction): against float64 cos of the
    float32 argument, the accuracy class of torch's float32 cos."""
    from www2023tiger_amd import hip_ops
    rs = np.random.RandomState(1)
    ts = np.concatenate([rs.uniform(3e6, 4e6, 200), rs.uniform(1e7, 1e9, 200), rs.uniform(1e9, 1e12, 200),
                         -rs.uniform(3e6, 1e10, 200), [3.0e6, 3.0000002e6, 2.9999998e6]]).astype(np.float32)
    freq = np.array([1.0, 0.73, 0.5, 1.0], dtype=np.float32)
    phase = np.array([0.0, 0.3, -0.2, 1.5], dtype=np.float32)
    out = hip_ops.time_encode(torch.from_numpy(ts).to(dev()), torch.from_numpy(freq).to(dev()),
                              torch.from_numpy(phase).to(dev())).cpu().numpy()
    arg = (ts[:, None] * freq[None, :]).astype(np.float32) + phase[None, :]   # float32 product, then the phase
    ref = np.cos(arg.astype(np.float32).astype(np.float64))
    assert np.abs(out - ref).max() < 3e-7
    tref = torch.cos(torch.from_numpy(arg.astype(np.float32))).numpy()
    assert np.abs(out - tref).max() < 5e-7


@pytest.mark.parametrize('name', ['static_ll_d16', 'seq_lr_d8', 'seq_rr_d8_nofeat', 'static_ll_d172'])
def test_fused_attention_weights_match_reference(name):
    """tg_attn_fuse (q+g and v+out+fc1 products pre-multiplied): embeddings against the reference fixtures
    and against the unfused path, including centres without neighbours (the masked constant)."""
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    plain, _, _ = build_hip_model(z, cfg)
    model.fuse_attention()
    assert model.model_struct().attn_fused
    B = cfg['B']
    stop = cfg['restart_at'] if cfg.get('restart_at', -1) >= 0 else 99  # the fixtures restart there; not replayed here
    for b in range(min(n_batches(z), 6, stop)):
        sl = slice(b * B, (b + 1) * B)
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        n = len(a[0])
        h = model.stream_step(*a).h[:2 * n].cpu().numpy()
        h0 = plain.stream_step(*a).h[:2 * n].cpu().numpy()
        assert rel_err(h, z[f'b{b}_h_left']) < TOL, b
        assert rel_err(h, h0) < 2e-5, b
    assert rel_err(model.left_memory.vals.cpu().numpy(), plain.left_memory.vals.cpu().numpy()) < 2e-5
    assert rel_err(model.right_memory.vals.cpu().numpy(), plain.right_memory.vals.cpu().numpy()) < 2e-5
    model.fuse_attention(False)
    assert not model.model_struct().attn_fused


@pytest.mark.parametrize('case', ['sampler_fixture', 'zipf_300k', 'tiny', 'one_node_heavy'])
def test_device_tcsr_build_equals_host_build(case):
    """tg_tcsr_build_device (stable radix sort on the owner id) against tg_tcsr_build_host: identical
    indptr / ts / nbr / eid+flag arrays, including self loops, id gaps and a node owning most entries."""
    from www2023tiger_amd.data.graph import Graph
    rs = np.random.RandomState(7)
    if case == 'sampler_fixture':
        z = load('sampler')
        src, dst, ts, eids = z['src'], z['dst'], z['ts'], z['eids']
    elif case == 'zipf_300k':
        import bench
        st = bench.make_stream(200000, 5000, 300000, 1.0e6, seed=2, d_e=4, with_efeats=False)
        src, dst, ts, eids = st['src'], st['dst'], st['ts'], st['eids']
    elif case == 'tiny':
        src, dst = np.array([3, 3, 1]), np.array([3, 2, 3])
        ts, eids = np.array([1.0, 1.0, 2.0]), np.array([1, 2, 3])
    else:
        E = 50000
        src = np.where(rs.uniform(size=E) < 0.9, 7, rs.randint(1, 4000, E))
        dst = rs.randint(1, 70000, E)
        ts, eids = np.sort(rs.uniform(0, 1e5, E)), np.arange(1, E + 1)
    g = Graph.from_arrays(src, dst, ts, eids, strategy='recent_edges', seed=0, device=dev())
    assert g._time_ordered
    got = [t.cpu().numpy() for t in g._tensors()]          # built on the GPU
    want = g._host_tcsr()                                   # built by the host routine
    for a, b, nm in zip(got, want, ('indptr', 'ts', 'nbr', 'eid')):
        np.testing.assert_array_equal(a, b, err_msg=nm)
    # an unsorted stream falls back to the host builder (per-node stable sort by time)
    perm = rs.permutation(len(src))
    g2 = Graph.from_arrays(src[perm], dst[perm], ts[perm], eids[perm], strategy='recent_edges', seed=0, device=dev())
    assert not g2._time_ordered or len(src) < 2
    g2._tensors()
