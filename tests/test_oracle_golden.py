"""The oracle (oracle/tiger_oracle.py) against the golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only.

Tolerances: integer outputs bit-exact; float32 state/embeddings 1e-5 relative
per step (same torch build produced both, only operation order differs)."""
import numpy as np
import pytest

from _util import MODEL_FIXTURES, TWO_LAYER_FIXTURES, fixture_params, fixture_tables, load, n_batches, parse_cfg, rel_err
from oracle import tiger_oracle as O

TOL = 2e-5


@pytest.fixture(scope='module')
def samp():
    return load('sampler')


def _graph(z, strategy, seed=3):
    return O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'], strategy=strategy, seed=seed)


@pytest.mark.parametrize('strategy', ['recent_edges', 'recent_nodes'])
@pytest.mark.parametrize('K', [1, 5, 10, 40])
def test_sampler_bit_exact(samp, strategy, K):
    g = _graph(samp, strategy)
    assert g.num_node == int(samp['num_node'])
    res = g.sample_temporal_neighbor(samp['q_nids'], samp['q_ts'], K)
    for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
        exp = samp[f'{strategy}_K{K}_{nm}']
        assert a.dtype == exp.dtype
        np.testing.assert_array_equal(a, exp)


@pytest.mark.parametrize('K', [5, 10])
def test_sampler_uniform_stream(samp, K):
    g = _graph(samp, 'uniform')
    for rep in range(2):
        res = g.sample_temporal_neighbor(samp['q_nids'], samp['q_ts'], K)
        for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
            np.testing.assert_array_equal(a, samp[f'uniform_K{K}_rep{rep}_{nm}'])


def test_history_float32_queries(samp):
    g = _graph(samp, 'recent_edges')
    res = g.get_history(samp['q_nids'], samp['q_ts'].astype(np.float32), 8)
    for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
        np.testing.assert_array_equal(a, samp[f'hist32_H8_{nm}'])


@pytest.mark.parametrize('i', [0, 1, 2])
def test_select_latest(samp, i):
    u, idx = O.select_latest_nids(samp[f'sel{i}_ids'], samp[f'sel{i}_ts'])
    np.testing.assert_array_equal(u, samp[f'sel{i}_unique'])
    np.testing.assert_array_equal(idx, samp[f'sel{i}_index'])


def test_select_latest_survey_example():
    u, idx = O.select_latest_nids(np.array([5, 3, 5, 3, 7]), np.array([1., 2., 1., 2., 0.]))
    assert u.tolist() == [3, 5, 7] and idx.tolist() == [1, 0, 4]


def test_anonymized_reindex(samp):
    np.testing.assert_array_equal(O.anonymized_reindex(samp['anon_in']), samp['anon_out'])
    np.testing.assert_array_equal(O.anonymized_reindex(samp['anon2_in']), samp['anon2_out'])


def build_oracle(z, cfg):
    n_nodes, nfeats, efeats = fixture_tables(z, cfg)
    g = O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'], strategy='recent_edges', seed=0)
    assert g.num_node == n_nodes
    return O.OracleTIGER(fixture_params(z, cfg), g, n_nodes=n_nodes, dim=cfg['d'], nfeats=nfeats, efeats=efeats,
                         n_neighbors=cfg['K'], msg_src=cfg['msg_src'], upd_src=cfg['upd_src'],
                         restarter=cfg['restarter'], hist_len=cfg.get('H'), tsfm=cfg.get('tsfm', 'id'),
                         upd_fn=cfg.get('upd_fn', 'gru'), hit_type=cfg.get('hit', 'bin'))


def check_state(m, z, tag):
    assert rel_err(m.left_vals.numpy(), z[f'{tag}_left_vals']) < TOL
    assert rel_err(m.right_vals.numpy(), z[f'{tag}_right_vals']) < TOL
    np.testing.assert_array_equal(m.left_ts.numpy(), z[f'{tag}_left_ts'])
    np.testing.assert_array_equal(m.right_ts.numpy(), z[f'{tag}_right_ts'])
    has = np.nonzero(m.has_msg)[0]
    np.testing.assert_array_equal(has, z[f'{tag}_has_msg'])
    assert rel_err(m.msg_vals.numpy()[has], z[f'{tag}_msg_vals']) < TOL
    np.testing.assert_array_equal(m.msg_ts.numpy()[has], z[f'{tag}_msg_ts'])


@pytest.mark.parametrize('name', MODEL_FIXTURES + TWO_LAYER_FIXTURES)
def test_stream_matches_reference(name):
    z = load(name)
    cfg = parse_cfg(z)
    m = build_oracle(z, cfg)
    B = cfg['B']
    restarting, uptodate = False, set()
    for b in range(n_batches(z)):
        sl = slice(b * B, min((b + 1) * B, len(z['src'])))
        src, dst, neg, ts, eids = (z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids'))
        cg = O.collate(m.graph, src, dst, neg, ts, cfg['K'], cfg['restarter'], cfg.get('H'), n_layers=cfg.get('L', 1))
        tag = f'b{b}'
        # collation is integer work: bit-exact
        for k in ('l1_nids', 'l1_eids', 'l1_ts', 'involved', 'rd_index', 'rd_nids', 'rd_ts',
                  'src_hits', 'dst_hits', 'neg_src_hits', 'neg_dst_hits') + (
                      ('hop2_nids', 'hop2_eids', 'hop2_ts') if cfg.get('L', 1) == 2 else ()):
            np.testing.assert_array_equal(cg[k], z[f'{tag}_{k}'], err_msg=k)
        if cfg['restarter'] == 'seq':
            for k in ('rd_hist_nids', 'rd_anon', 'rd_hist_eids', 'rd_hist_ts', 'rd_hist_dirs'):
                np.testing.assert_array_equal(cg[k], z[f'{tag}_{k}'], err_msg=k)
        else:
            np.testing.assert_array_equal(cg['rd_prev_ts'], z[f'{tag}_rd_prev_ts'])
        # lazy restart bookkeeping as train_self_supervised.py:152-163
        if b == cfg.get('restart_at', -1):
            restarting, uptodate = True, set()
            m.clear_msgs()
        if restarting:
            r = np.array(sorted(set(cg['involved'].tolist()) - uptodate), dtype=np.int64)
            np.testing.assert_array_equal(r, z[f'{tag}_restart_nids'])
            r_ts = np.full(len(r), np.float32(ts.min()), dtype=np.float32)
            if len(r):
                hl, hr, pt = m.restarter_forward(r, r_ts)
                assert rel_err(hl.numpy(), z[f'{tag}_restart_h_left']) < TOL
                assert rel_err(hr.numpy(), z[f'{tag}_restart_h_right']) < TOL
                np.testing.assert_array_equal(pt.numpy(), z[f'{tag}_restart_prev_ts'])
            m.restart(r, r_ts)
            uptodate.update(r.tolist())
            check_state(m, z, f'{tag}_afterrestart')
        out = m.contrast_learning(src, dst, neg, ts, eids, cg)
        for k in ('h_left', 'pos_scores', 'neg_scores', 'h_prev_left', 'h_prev_right'):
            assert rel_err(out[k].numpy(), z[f'{tag}_{k}']) < TOL, (b, k)
        assert abs(float(out['loss']) - float(z[f'{tag}_loss'])) < 1e-5
        # mutual-learning surrogate on the collated restart data (tiger.py:576-590)
        pos = np.concatenate([src, dst])
        idx = cg['rd_index']
        ts2 = np.tile(ts, 2).astype(np.float32)
        sl_, sr_, spt = m.restarter_forward(pos[idx], ts2[idx], cg)
        assert rel_err(sl_.numpy(), z[f'{tag}_sur_left']) < TOL
        assert rel_err(sr_.numpy(), z[f'{tag}_sur_right']) < TOL
        np.testing.assert_array_equal(np.asarray(spt).reshape(z[f'{tag}_sur_prev_ts'].shape), z[f'{tag}_sur_prev_ts'])
        if f'{tag}_left_vals' in z.files:
            check_state(m, z, tag)
    m.flush_msg()
    check_state(m, z, 'flushed')


# --------------------------------------------------------------------------- training tail
TRAIN_FIXTURES = ['train_seq_lr_d8', 'train_static_ll_d16', 'train_contrast_rr_d8', 'train_mlp_merge_d8',
                  'train_linear_gru_d8', 'train_seq_lr_d8_zeronf']
TRAIN_FIXTURES_L2 = ['train_static_lr_d8_L2', 'train_contrast_ll_d16_L2']  # --n_layers 2


def grad_err(a, b):
    """max |a-b| relative to the largest entry of the reference gradient (floor 1e-3)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-3, np.abs(b).max()))


def oracle_train_batch(m, z, cfg, b, state):
    """One iteration of train_self_supervised.py:143-171 on the oracle."""
    B = cfg['B']
    lo, hi = b * B, min((b + 1) * B, len(z['src']))
    a = [z[k][lo:hi] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
    cg = O.collate(m.graph, a[0], a[1], a[2], a[3], cfg['K'], cfg['restarter'], hist_len=cfg.get('H'), n_layers=cfg.get('L', 1))
    if b == cfg.get('restart_at', -1):
        state['restarting'], state['uptodate'] = True, set()
        m.clear_msgs()
    if state.get('restarting'):
        r_nodes = np.array(sorted(set(cg['involved'].tolist()) - state['uptodate']), dtype=np.int64)
        m.restart(r_nodes, np.full(len(r_nodes), np.float32(a[3]).min(), dtype=np.float32))
        state['uptodate'].update(r_nodes.tolist())
    return m.train_step(*a, cg, lr=cfg['lr'], mutual_coef=cfg['mutual_coef'],
                        contrast_only=bool(cfg.get('contrast_only', 0)))


@pytest.mark.parametrize('name', TRAIN_FIXTURES + TRAIN_FIXTURES_L2)
def test_training_matches_reference(name):
    z = load(name)
    cfg = parse_cfg(z)
    m = build_oracle(z, cfg)
    state = {}
    for b in range(cfg['n_batches']):
        c, ml, grads = oracle_train_batch(m, z, cfg, b, state)
        assert abs(c - float(z[f'b{b}_contrast_loss'])) < 2e-5 * max(1.0, abs(c)), b
        assert abs(ml - float(z[f'b{b}_mutual_loss'])) < 2e-5 * max(1.0, abs(ml)), b
        if b in cfg['grad_batches']:
            for k, g in grads.items():
                assert grad_err(g.numpy(), z[f'b{b}_grad.{k}']) < 1e-4, (b, k)
    for k, v in m.p.items():
        assert rel_err(v.detach().numpy(), z[f"final.{k}"]) < 1e-3, k  # Adam amplifies ulp-level gradient noise
    check_state(m, z, 'final')


# ------------------------------------------------------------------------------ vectorised forms of the oracle loops
def test_vectorised_sampler_and_dedup_equal_the_loops(samp, monkeypatch):
    """The full-size parity cases use numpy-vectorised forms of the per-query sampler loop and of the
    select_latest loop; they must agree with the loops bit for bit (and with the reference's vectors)."""
    g = _graph(samp, 'recent_edges')
    rs = np.random.RandomState(11)
    q = np.concatenate([samp['q_nids'], rs.randint(0, g.num_node, 5000)])
    t = np.concatenate([samp['q_ts'], rs.choice(np.concatenate([samp['ts'], samp['ts'] + 0.5, [-1.0, 1e12]]), 5000)])
    for K in (1, 10, 40):
        for tq in (t, t.astype(np.float32)):
            monkeypatch.setattr(O, 'VECTORISE_FROM', 10 ** 9)
            loop = g.sample_temporal_neighbor(q, tq, K)
            monkeypatch.setattr(O, 'VECTORISE_FROM', 0)
            vec = g.sample_temporal_neighbor(q, tq, K)
            for a, b in zip(loop, vec):
                assert a.dtype == b.dtype
                np.testing.assert_array_equal(a, b)
    res = g.sample_temporal_neighbor(samp['q_nids'], samp['q_ts'], 10)   # still VECTORISE_FROM == 0
    for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
        np.testing.assert_array_equal(a, samp[f'recent_edges_K10_{nm}'])
    ids = rs.randint(0, 300, 20000)
    for ts in (np.floor(rs.uniform(0, 40, 20000)), np.floor(rs.uniform(0, 40, 20000)).astype(np.float32)):
        monkeypatch.setattr(O, 'VECTORISE_FROM', 10 ** 9)
        u0, i0 = O.select_latest_nids(ids, ts)
        monkeypatch.setattr(O, 'VECTORISE_FROM', 0)
        u1, i1 = O.select_latest_nids(ids, ts)
        np.testing.assert_array_equal(u0, u1)
        np.testing.assert_array_equal(i0, i1)
    for i in range(3):
        u, idx = O.select_latest_nids(samp[f'sel{i}_ids'], samp[f'sel{i}_ts'])
        np.testing.assert_array_equal(u, samp[f'sel{i}_unique'])
        np.testing.assert_array_equal(idx, samp[f'sel{i}_index'])
    empty = O.OracleGraph(np.array([1]), np.array([2]), np.array([5.0]), np.array([1]), max_node_id=9)
    out = empty.sample_temporal_neighbor(np.array([0, 1, 9]), np.array([1.0, 9.0, 9.0]), 3)
    assert out[0].tolist() == [[0, 0, 0], [0, 0, 2], [0, 0, 0]]
