"""Evaluation harness (www2023tiger_amd/eval_utils.py) against the AP / AUC the reference's own
tiger/eval_utils.py produced on the same streams (tests/golden/eval_*.npz), and tg_ap_auc
against sklearn."""
import numpy as np
import pytest
import torch

from _util import load, parse_cfg, rel_err
from test_hip_parity import build_hip_model, dev

pytestmark = pytest.mark.gpu


def test_ap_auc_matches_sklearn():
    from sklearn.metrics import average_precision_score, roc_auc_score
    from www2023tiger_amd.eval_utils import ap_auc_windows
    rs = np.random.RandomState(0)
    n, w = 1000, 200
    pos = rs.uniform(0, 1, n).astype(np.float32)
    neg = rs.uniform(0, 1, n).astype(np.float32)
    pos[::7] = np.float32(1.0)          # saturated sigmoids: ties between and within the classes
    neg[::11] = np.float32(1.0)
    neg[5:40] = pos[5:40]
    neg[300:310] = np.nan                # dropped from their window
    ap, auc, bad = ap_auc_windows(torch.from_numpy(pos).to(dev()), torch.from_numpy(neg).to(dev()), w)
    assert int(bad.item()) == 10
    for i in range(n // w):
        sc = np.concatenate([pos[i * w:(i + 1) * w], neg[i * w:(i + 1) * w]])
        lab = np.concatenate([np.ones(w), np.zeros(w)])
        ok = np.isfinite(sc)
        assert abs(float(ap[i]) - average_precision_score(lab[ok], sc[ok])) < 1e-12, i
        assert abs(float(auc[i]) - roc_auc_score(lab[ok], sc[ok])) < 1e-12, i
    # ragged tail window
    ap2, auc2, _ = ap_auc_windows(torch.from_numpy(pos[:250]).to(dev()), torch.from_numpy(neg[:250]).to(dev()), w)
    sc = np.concatenate([pos[200:250], neg[200:250]])
    assert abs(float(ap2[1]) - average_precision_score(np.r_[np.ones(50), np.zeros(50)], sc)) < 1e-12


@pytest.mark.parametrize('name', ['eval_seq_lr_d8', 'eval_static_ll_d16'])
def test_eval_harness_matches_reference(name):
    from torch.utils.data import DataLoader
    from www2023tiger_amd.data.data_loader import InteractionData
    from www2023tiger_amd.eval_utils import eval_edge_prediction, warmup
    z = load(name)
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    data = InteractionData(z['src'], z['dst'], z['ts'], z['eids'], np.zeros(len(z['src']), dtype=np.int64), seed=0,
                           eval=True)
    np.testing.assert_array_equal(data.neg_dst, z['neg'])  # RandEdgeSampler.pre_sample_neg_dsts stream
    mk = lambda lo, hi: DataLoader(data.get_subset(lo, hi), batch_size=cfg['B'], shuffle=False, collate_fn=coll)
    n_warm, n_val = cfg['n_warm'], cfg['n_val']
    tol = 2e-4  # scores are float32 to 1e-4; a window's AP/AUC moves only when a near-tie flips
    model.reset()
    ap, auc = eval_edge_prediction(model, mk(0, n_val), dev(), restart_mode=False, mean_over_n_samples=cfg['chunk'])
    assert abs(ap - float(z['stream_ap'])) < tol and abs(auc - float(z['stream_auc'])) < tol
    model.reset()
    up = warmup(model, mk(0, n_warm), dev())
    np.testing.assert_array_equal(np.array(sorted(up), dtype=np.int64), z['warm_uptodate'])
    state = model.save_memory_state()
    ap, auc = eval_edge_prediction(model, mk(n_warm, n_warm + n_val), dev(), restart_mode=True,
                                   uptodate_nodes=set(up), mean_over_n_samples=cfg['chunk'])
    assert abs(ap - float(z['restart_ap'])) < tol and abs(auc - float(z['restart_auc'])) < tol
    model.load_memory_state(state)
    ap, auc = eval_edge_prediction(model, mk(n_warm, n_warm + n_val), dev(), restart_mode=True, uptodate_nodes=set(up))
    assert abs(ap - float(z['restart200_ap'])) < tol and abs(auc - float(z['restart200_auc'])) < tol
    assert rel_err(model.left_memory.vals.cpu().numpy(), z['final_left_vals']) < 1e-4
    assert rel_err(model.right_memory.vals.cpu().numpy(), z['final_right_vals']) < 1e-4


def test_end_to_end_recipe_on_toy_jodie_files(tmp_path):
    """init_data -> init_model -> train (torch Adam, lazy restarts, mutual learning) -> validate with
    snapshots -> checkpoint round trip -> test: the reference's script flow on this package's API."""
    import os
    import sys
    from test_input_side import write_files
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'examples'))
    import link_prediction as lp
    z = load('input_side')
    write_files(str(tmp_path), 'toy', z, with_feats=False)  # LastFM-style: no feature files, width from --dim
    ckpt = str(tmp_path / 'model.pt')
    out, model = lp.run('toy', str(tmp_path), seed=0, n_epochs=2, bs=100, lr=1e-3, dim=8, n_neighbors=4, hist_len=6,
                        restarter_type='seq', restart_prob=0.2, warmup_steps=100, ckpt_path=ckpt)
    e0, e1 = out['epochs']
    assert all(np.isfinite(list(e.values())).all() for e in out['epochs'])
    assert e1['contrast'] < e0['contrast']           # it learns
    assert 0.0 <= out['test_ap'] <= 1.0 and 0.0 <= out['ind_test_auc'] <= 1.0
    # the checkpoint carries parameters and both memories (with the alias keys); the mailbox buffers are
    # non-persistent in the reference (memory.py:62-67) and here
    sd = torch.load(ckpt)
    for k in ('left_memory.vals', 'right_memory.update_ts', 'msg_memory.vals', 'upd_memory.vals',
              'score_fn.fc1.weight', 'restarter_fn.mha_fn.in_proj_weight'):
        assert k in sd, k
    assert not any(k.startswith('msg_store.') for k in sd)


def test_recipe_when_the_training_split_lacks_the_highest_node_id(tmp_path):
    """Real JODIE splits: items first seen after the validation time (and the hidden 10 % of the nodes) are missing
    from the training split, so Graph.from_data(train_data) alone would cover fewer ids than the model's tables.
    init_data builds both graphs over the id space of the full data; the loop must run end to end."""
    import os
    import sys
    from test_input_side import write_files
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'examples'))
    import link_prediction as lp
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.init_utils import init_data
    z0 = load('input_side')
    z = {k: z0[k] for k in ('src', 'dst', 'ts', 'labels')}
    for _ in range(3):  # three late events on brand-new items: the highest ids occur in the test period only
        z['src'] = np.append(z['src'], z['src'][-1])
        z['dst'] = np.append(z['dst'], z['dst'].max() + 1)
        z['ts'] = np.append(z['ts'], z['ts'][-1] + 1.0)
        z['labels'] = np.append(z['labels'], 0)
    write_files(str(tmp_path), 'late', z, with_feats=False)
    basic, (train_graph, full_graph), _ = init_data('late', str(tmp_path), 0, bs=100, warmup_steps=0, subset=1.0,
                                                    strategy='recent_edges', n_layers=1, n_neighbors=4,
                                                    restarter_type='static', hist_len=6, device=dev())
    train_data, full_data = basic[3], basic[2]
    assert max(train_data.src.max(), train_data.dst.max()) < max(full_data.src.max(), full_data.dst.max())
    assert train_graph.num_node == full_graph.num_node == int(max(full_data.src.max(), full_data.dst.max())) + 1
    out, model = lp.run('late', str(tmp_path), seed=0, n_epochs=1, bs=100, lr=1e-3, dim=8, n_neighbors=4, hist_len=6,
                        restarter_type='static', restart_prob=0.1, warmup_steps=50, ckpt_path=str(tmp_path / 'm.pt'))
    assert np.isfinite(list(out['epochs'][0].values())).all() and 0.0 <= out['test_ap'] <= 1.0
    # a graph over fewer ids than the model is refused with a pointer to the fix, not with an opaque status code
    small = Graph.from_data(train_data, strategy='recent_edges', seed=0, device=dev())
    assert small.num_node < model.n_nodes
    with pytest.raises(ValueError, match='max_node_id'):
        model.check_graph(small)


def test_fused_eval_samples_from_the_collators_graph():
    """Warm-up batches are collated on the TRAINING graph while model.graph is the full graph
    (train_self_supervised.py:175-183): the one-call evaluation must embed with the collator's
    neighbourhoods, exactly as the operator path (which consumes the collated layers) does."""
    from www2023tiger_amd.data.data_loader import GraphCollator
    from www2023tiger_amd.data.graph import Graph
    z = load('eval_seq_lr_d8')
    cfg = parse_cfg(z)
    model, full_graph, _ = build_hip_model(z, cfg, dropout=0.0)
    keep = np.ones(len(z['src']), dtype=bool)
    keep[::3] = False  # a "training" graph that misses a third of the events
    train_graph = Graph.from_arrays(z['src'][keep], z['dst'][keep], z['ts'][keep], z['eids'][keep],
                                    strategy='recent_edges', seed=0, max_node_id=int(z['n_nodes']) - 1, device=dev())
    coll = GraphCollator(train_graph, cfg['K'], 1, restarter=cfg['restarter'], hist_len=cfg.get('H'))
    ref, _, _ = build_hip_model(z, cfg, dropout=0.0)
    ref._fused_eval_ok = lambda *a: False  # operator path: uses the collated layers as they are
    model.eval(); ref.eval()
    assert model.graph is not train_graph
    B = cfg['B']
    for b in range(6, 10):
        sl = slice(b * B, (b + 1) * B)
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        s, d_, n_, t, e, _, cg = coll.collate_arrays(*a)
        out = model.contrast_learning(s, d_, n_, t, e, cg)
        out_ref = ref.contrast_learning(s, d_, n_, t, e, cg)
        for x, y in zip(out[1:4], out_ref[1:4]):
            assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < 1e-5, b


def test_batch_loader_equals_dataloader():
    """BatchLoader (arrays + one native negative-sampling call per batch) against torch's DataLoader over the same
    dataset and collator: identical batches, training-mode negatives included."""
    from torch.utils.data import DataLoader
    from www2023tiger_amd.data.data_loader import BatchLoader, ChunkSampler, InteractionData
    z = load('eval_seq_lr_d8')
    cfg = parse_cfg(z)
    _, _, coll = build_hip_model(z, cfg, dropout=0.0)
    lab = np.zeros(len(z['src']), dtype=np.int64)
    mk = lambda: InteractionData(z['src'], z['dst'], z['ts'], z['eids'], lab, seed=9, eval=False)
    ref = DataLoader(mk(), batch_size=64, shuffle=False, collate_fn=coll)
    got = BatchLoader(mk(), 64, coll)
    assert len(ref) == len(got)
    for a, b in zip(ref, got):
        for x, y in zip(a[:5], b[:5]):
            assert torch.equal(x, y)
        np.testing.assert_array_equal(a[6].np_computation_graph_nodes, b[6].np_computation_graph_nodes)
    cs = ChunkSampler(len(lab), rank=1, world_size=2, bs=64, seed=3)
    ref = DataLoader(mk(), batch_size=64, sampler=cs, collate_fn=coll)
    got = BatchLoader(mk(), 64, coll, sampler=cs)
    assert len(ref) == len(got)
    for a, b in zip(ref, got):
        assert torch.equal(a[0], b[0]) and torch.equal(a[4], b[4])


def test_deferred_invariant_check_still_raises():
    """The one-call evaluation step reads its invariant word back asynchronously (no stall per batch): replaying
    a batch - events older than what the memories have seen, tiger.py:437-438 - must still raise the reference's
    ValueError, at the next batch or at the end of the harness, and the harness must be usable afterwards."""
    from torch.utils.data import DataLoader
    from www2023tiger_amd.data.data_loader import InteractionData
    from www2023tiger_amd.eval_utils import eval_edge_prediction
    z = load('eval_static_ll_d16')
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    data = InteractionData(z['src'], z['dst'], z['ts'], z['eids'], np.zeros(len(z['src']), dtype=np.int64), seed=0,
                           eval=True)
    B = cfg['B']
    dl = DataLoader(data.get_subset(0, 2 * B), batch_size=B, shuffle=False, collate_fn=coll)
    model.reset()
    eval_edge_prediction(model, dl, dev(), restart_mode=False)
    with pytest.raises(ValueError):  # the same two batches again: the first replayed batch violates the order
        eval_edge_prediction(model, dl, dev(), restart_mode=False)
    model.reset()
    ap, auc = eval_edge_prediction(model, dl, dev(), restart_mode=False)  # clean again after a reset
    assert np.isfinite(ap) and np.isfinite(auc)


@pytest.mark.parametrize('eval_split', [True, False])
def test_resident_eval_equals_the_per_batch_loop(eval_split, monkeypatch):
    """eval_edge_prediction over a BatchLoader takes the resident-stream form (columns uploaded once, one call per
    batch at a device-side offset, scores left in place).  With the model's own forms (TG_EVAL_STREAM=0) AP / AUC and
    the memories after the pass equal the literal per-batch loop's bit for bit; by default the pass streams with eager
    updates and pre-multiplied weights (put back afterwards), which agrees to float32 rounding.  Full batches plus a
    ragged tail, an evaluation split (fixed negatives) and a training split (negatives drawn from the sampler's
    RandomState stream, on the device)."""
    from www2023tiger_amd import eval_utils
    from www2023tiger_amd.data.data_loader import BatchLoader, InteractionData
    z = load('eval_static_ll_d16')
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    B = cfg['B']
    n = 12 * B + B // 3
    mk = lambda: BatchLoader(InteractionData(z['src'][:n], z['dst'][:n], z['ts'][:n], z['eids'][:n], np.zeros(n, dtype=np.int64),
                                             seed=5, eval=eval_split), B, coll)
    taken = []
    real = eval_utils._eval_resident
    monkeypatch.setattr(eval_utils, '_eval_resident', lambda *a: (taken.append(1), real(*a))[1])
    out = {}
    for form, env in (('loop', dict(TG_EVAL_RESIDENT='0')), ('resident', dict(TG_EVAL_RESIDENT='1', TG_EVAL_STREAM='0')),
                      ('stream', dict(TG_EVAL_RESIDENT='1', TG_EVAL_STREAM='1'))):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model.reset()
        res = eval_utils.eval_edge_prediction(model, mk(), dev(), restart_mode=False, mean_over_n_samples=cfg['chunk'])
        assert model._pending is None and model._fused is None  # the switches of the streaming form are put back
        out[form] = (res, model.left_memory.vals.clone(), model.right_memory.vals.clone(), model.left_memory.update_ts.clone(),
                     model.msg_store.node_msg_vals.clone())
    assert len(taken) == 2
    assert out['loop'][0] == out['resident'][0]
    for a, b in zip(out['loop'][1:], out['resident'][1:]):
        assert torch.equal(a, b)
    assert abs(out['loop'][0][0] - out['stream'][0][0]) < 2e-4 and abs(out['loop'][0][1] - out['stream'][0][1]) < 2e-4
    assert torch.equal(out['loop'][3], out['stream'][3])
    for a, b in zip(out['loop'][1:], out['stream'][1:]):
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5


def test_eval_loop_on_a_model_that_streams_with_eager_updates():
    """contrast_learning under no_grad on a model with eager_updates() + fuse_attention(): the one-call evaluation
    step takes the streaming step's eager forward and keeps the per-node tables current (no rebuild per batch); scores
    and state agree with the lazy, unfused model, and a streaming step afterwards finds consistent tables."""
    from www2023tiger_amd.data.data_loader import BatchLoader, InteractionData
    z = load('eval_static_ll_d16')
    cfg = parse_cfg(z)
    B = cfg['B']
    n = 4 * B
    outs = []
    for eager in (False, True):
        model, _, coll = build_hip_model(z, cfg, dropout=0.0)
        model.eval()
        if eager:
            model.eager_updates(True).fuse_attention(True)
        model.reset()
        dl = BatchLoader(InteractionData(z['src'][:n], z['dst'][:n], z['ts'][:n], z['eids'][:n], np.zeros(n, dtype=np.int64),
                                         seed=0, eval=True), B, coll)
        sc = []
        for k, (s, d_, ng, t, e, _, cg) in enumerate(dl):
            if k == 3:  # the last batch as a plain streaming step: reads the tables the evaluation steps kept
                buf = model.stream_step(s, d_, ng, cg.ts64, e)
                sc.append(buf.h.clone())
            else:
                out = model.contrast_learning(s, d_, ng, t, e, cg)
                sc.append(torch.cat([out[2], out[3]]))
        model._poll_train_errors()
        outs.append((sc, model.left_memory.vals.clone(), model.right_memory.vals.clone()))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5


def test_resident_eval_raises_the_invariant_errors():
    from www2023tiger_amd.data.data_loader import BatchLoader, InteractionData
    from www2023tiger_amd.eval_utils import eval_edge_prediction
    z = load('eval_static_ll_d16')
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    B = cfg['B']
    n = 2 * B
    dl = BatchLoader(InteractionData(z['src'][:n], z['dst'][:n], z['ts'][:n], z['eids'][:n], np.zeros(n, dtype=np.int64), seed=0,
                                     eval=True), B, coll)
    model.reset()
    eval_edge_prediction(model, dl, dev(), restart_mode=False)
    with pytest.raises(ValueError):  # the same batches again: events older than what the memories have seen
        eval_edge_prediction(model, dl, dev(), restart_mode=False)
    model.reset()
    ap, auc = eval_edge_prediction(model, dl, dev(), restart_mode=False)
    assert np.isfinite(ap) and np.isfinite(auc)


@pytest.mark.parametrize('name', ['eval_seq_lr_d8', 'eval_static_ll_d16'])
@pytest.mark.parametrize('stream', ['0', '1', '1g', '1s'],
                         ids=['own_forms', 'eager_fused', 'eager_fused_restart_graph', 'eager_fused_one_stream'])
def test_resident_eval_in_restart_mode_equals_the_per_batch_loop(name, stream, monkeypatch):
    """restart_mode=True (the reference's default recipe: --restart_prob 0.01): the lazy restart of eval_utils.py:37-42
    with its bookkeeping on the device (involved & ~uptodate listed by a collate-only pass, ONE count read back per
    batch) against the literal loop with Python sets: AP / AUC, the up-to-date set handed back in place and the state."""
    from www2023tiger_amd import eval_utils
    from www2023tiger_amd.data.data_loader import BatchLoader, InteractionData
    z = load(name)
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    B = cfg['B']
    n_warm, n = cfg['n_warm'], 5 * B + B // 2
    data = InteractionData(z['src'], z['dst'], z['ts'], z['eids'], np.zeros(len(z['src']), dtype=np.int64), seed=0, eval=True)
    mk = lambda lo, hi: BatchLoader(data.get_subset(lo, hi), B, coll)
    if stream == '1g':  # the restarts of batches with few nodes as a replayed graph (device-side count; off by default)
        monkeypatch.setenv('TG_EVAL_RESTART_RUN', '0')
        monkeypatch.setenv('TG_EVAL_RESTART_GRAPH', '1')
        stream = '1'
    if stream == '1s':  # the host-sequenced pipeline with the restarter's forward on the steps' stream
        monkeypatch.setenv('TG_EVAL_RESTART_RUN', '0')
        monkeypatch.setenv('TG_EVAL_RESTART_OVERLAP', '0')
        stream = '1'
    monkeypatch.setenv('TG_EVAL_STREAM', stream)
    out = {}
    for form in ('0', '1'):
        monkeypatch.setenv('TG_EVAL_RESIDENT', form)
        model.reset()
        up = eval_utils.warmup(model, mk(0, n_warm), dev())
        n_up = len(up)
        res = eval_utils.eval_edge_prediction(model, mk(n_warm, n_warm + n), dev(), restart_mode=True, uptodate_nodes=up,
                                              mean_over_n_samples=cfg['chunk'])
        assert len(up) > n_up  # handed back in place
        out[form] = (res, sorted(up), model.left_memory.vals.clone(), model.right_memory.vals.clone(),
                     model.msg_store.node_msg_vals.clone(), model.msg_store.has_msg_mask().clone())
    assert out['0'][1] == out['1'][1]
    assert torch.equal(out['0'][5], out['1'][5])
    tol = 0.0 if stream == '0' else 2e-4
    assert abs(out['0'][0][0] - out['1'][0][0]) <= tol and abs(out['0'][0][1] - out['1'][0][1]) <= tol
    for a, b in zip(out['0'][2:5], out['1'][2:5]):
        if stream == '0':
            assert torch.equal(a, b)
        else:
            assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('restarter', ['seq', 'static'])
def test_restart_mode_eval_on_two_streams_equals_one_stream(restarter, monkeypatch):
    """The restart-mode pass with the SeqRestarter: passes and the restarter's forward on a side stream beside the previous
    batch's step - sequenced by the library (tg_eval_restart_run, the default) or by the host (eval_utils._RestartPipeline) -
    against the same calls on ONE stream and against the pass without the pipeline.  A Wikipedia-shaped stream where restarts
    go on for many batches (most nodes are met late), 60 batches + a ragged one: scores, the up-to-date set and the final
    state - bit for bit where the calls per batch are the same (only their order across streams differs), to rounding where
    one forward serves a group of batches."""
    import bench
    from www2023tiger_amd import eval_utils
    from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, InteractionData
    B, nb, d, K, H = 100, 60, 32, 10, 16
    n = nb * B + 37  # (+ a ragged last batch: a second pass, whose bitmap is handed over from the first)
    # the LAST n events of a stream twice as long are evaluated: the restarted nodes have histories (from the stream's first
    # event on every restart would meet an empty one - one compact row per node, no Q / K rows that matter)
    st = bench.make_stream(1500, 300, 2 * n, 2.0e5, seed=5, d_e=d)
    model, _ = bench.build_models(st, d, K, 'left', 'right', restarter=restarter, hist_len=H, dropout=0.1)
    if restarter == 'static':  # (the reference initialises the tables with zeros: trained values are what a restart is for)
        torch.manual_seed(3)
        with torch.no_grad():
            model.restarter_fn.left_emb.weight.normal_(0.0, 0.5)
            model.restarter_fn.right_emb.weight.normal_(0.0, 0.5)
    model.eval()
    coll = GraphCollator(model.graph, K, 1, restarter=restarter, hist_len=H)
    neg = np.random.RandomState(2).randint(1501, 1801, n)
    data = InteractionData(st['src'][n:], st['dst'][n:], st['ts'][n:], st['eids'][n:], np.zeros(n, dtype=np.int64), seed=0,
                           eval=True, neg_dst=neg)
    out = {}
    counts = []
    orig = eval_utils._RestartPipeline.restart
    monkeypatch.setattr(eval_utils._RestartPipeline, 'restart', lambda self, k: counts.append(orig(self, k)) or counts[-1])
    orig_run = eval_utils._RestartRun.run
    monkeypatch.setattr(eval_utils._RestartRun, 'run', lambda self, *a: (orig_run(self, *a), counts.extend(self.counts))[0])
    knobs = ('TG_EVAL_RESTART_RUN', 'TG_EVAL_RESTART_OVERLAP', 'TG_EVAL_RESTART_PIPELINE', 'TG_EVAL_RESTART_GROUP',
             'TG_EVAL_RESTART_INSTEP', 'TG_EVAL_PREFETCH')
    for form, env in (('run', {}), ('run1', dict(TG_EVAL_RESTART_GROUP='1')), ('run3', dict(TG_EVAL_RESTART_GROUP='3')),
                      ('chunks', dict(TG_EVAL_RESTART_GROUP='8')),  # (+ 64 nodes per forward: whole and partial lists per call)
                      ('run_nopf', dict(TG_EVAL_PREFETCH='0')),  # (no collate prefetch inside the groups)
                      ('two', dict(TG_EVAL_RESTART_RUN='0')),
                      ('one', dict(TG_EVAL_RESTART_RUN='0', TG_EVAL_RESTART_OVERLAP='0', TG_EVAL_RESTART_INSTEP='0')),
                      ('plain', dict(TG_EVAL_RESTART_PIPELINE='0'))):  # (static: 'two' and 'plain' are the in-step loop)
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if restarter == 'static' and 'TG_EVAL_RESTART_RUN' not in env:
            monkeypatch.setenv('TG_EVAL_RESTART_RUN', '2')  # (opt-in for the StaticRestarter: the in-step loop is its default)
        model.reset()
        up = set()
        del counts[:]
        monkeypatch.setattr(eval_utils._RestartRun, 'FWD_NODES', 64 if form == 'chunks' else 2048)
        res = eval_utils.eval_edge_prediction(model, BatchLoader(data, B, coll), dev(), restart_mode=True, uptodate_nodes=up,
                                              mean_over_n_samples=200)
        if form == 'chunks' and restarter == 'seq':
            assert max(counts) > 64  # a single list did exceed a forward's capacity
        if form in ('run', 'two') and not (form == 'two' and restarter == 'static'):
            assert sum(1 for c in counts if c) >= nb // 2  # restarts in most batches: the two streams did meet
        out[form] = (res, sorted(up), model.left_memory.vals.clone(), model.right_memory.vals.clone(),
                     model.left_memory.update_ts.clone(), model.msg_store.node_msg_vals.clone(),
                     model.msg_store.has_msg_mask().clone())
    # the same calls per batch, only their order across streams differs: bit for bit
    for other in ('two', 'one', 'plain'):
        assert out['run1'][0] == out[other][0] and out['run1'][1] == out[other][1], other
        for a, b in zip(out['run1'][2:], out[other][2:]):
            assert torch.equal(a, b), other
    # one forward per group / per chunk: other row counts, hence other blocks for its products (their sums in another
    # order) - the same lists, the same has-message bits, rows to rounding
    assert out['run'][0] == out['run_nopf'][0] and all(torch.equal(a, b) for a, b in zip(out['run'][2:], out['run_nopf'][2:]))
    for other in ('run', 'run3', 'chunks'):
        assert out['run1'][1] == out[other][1] and torch.equal(out['run1'][-1], out[other][-1]), other
        assert abs(out['run1'][0][0] - out[other][0][0]) <= 2e-4 and abs(out['run1'][0][1] - out[other][0][1]) <= 2e-4, other
        for a, b in zip(out['run1'][2:-1], out[other][2:-1]):
            assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5, other


@pytest.mark.parametrize('name,strategy', [('train_static_lr_d8_L2', 'recent_edges'), ('eval_static_ll_d16', 'recent_nodes'),
                                           ('eval_seq_lr_d8', 'recent_nodes'), ('seq_ll_d16_L2', 'recent_edges'),
                                           ('seq_ll_d16_L2', 'recent_nodes')])
@pytest.mark.parametrize('restart', [False, True], ids=['plain', 'restart_mode'])
def test_resident_eval_with_two_layers_and_recent_nodes(name, strategy, restart, monkeypatch):
    """The resident evaluation pass on the forms beside the default one: two attention layers (no per-node tables: the
    streaming form is the lean eager step without them) and the recent_nodes strategy (hit windows from one more
    recent-edges sampler launch) - against the literal per-batch loop."""
    from www2023tiger_amd import eval_utils
    from www2023tiger_amd.data.data_loader import BatchLoader, InteractionData
    z = load(name)
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, strategy=strategy, dropout=0.0)
    B = cfg['B']
    n = min(len(z['src']), 6 * B + B // 2)
    rs = np.random.RandomState(4)
    neg = rs.randint(int(z['dst'].min()), int(z['dst'].max()) + 1, len(z['src']))
    mk = lambda: BatchLoader(InteractionData(z['src'][:n], z['dst'][:n], z['ts'][:n], z['eids'][:n], np.zeros(n, dtype=np.int64),
                                             seed=1, eval=True, neg_dst=neg), B, coll)
    out = {}
    for form, env in (('loop', dict(TG_EVAL_RESIDENT='0')), ('resident', dict(TG_EVAL_RESIDENT='1', TG_EVAL_STREAM='0')),
                      ('stream', dict(TG_EVAL_RESIDENT='1', TG_EVAL_STREAM='1'))):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model.reset()
        up = set()
        assert (eval_utils._resident_plan(model, mk(), restart) is None) == (form == 'loop')  # (no silent fall-back to the loop)
        res = eval_utils.eval_edge_prediction(model, mk(), dev(), restart_mode=restart, uptodate_nodes=up, mean_over_n_samples=50)
        out[form] = (res, sorted(up), model.left_memory.vals.clone(), model.right_memory.vals.clone())
    assert out['loop'][0] == out['resident'][0] and out['loop'][1] == out['resident'][1] == out['stream'][1]
    for a, b in zip(out['loop'][2:], out['resident'][2:]):
        assert torch.equal(a, b)
    assert abs(out['loop'][0][0] - out['stream'][0][0]) < 5e-4 and abs(out['loop'][0][1] - out['stream'][0][1]) < 5e-4
    for a, b in zip(out['loop'][2:], out['stream'][2:]):
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5


def test_resident_eval_over_device_resident_columns_and_a_chunk_sampler():
    """The pass over a dataset whose columns already live on the GPU (InteractionData.to_device) and over a contiguous
    sub-range of it (ChunkSampler: the reference's time-chunk DDP sampler) - the same scores as over host columns."""
    from www2023tiger_amd import eval_utils
    from www2023tiger_amd.data.data_loader import BatchLoader, ChunkSampler, InteractionData
    z = load('eval_static_ll_d16')
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    B = cfg['B']
    n = 8 * B
    mk = lambda: InteractionData(z['src'][:n], z['dst'][:n], z['ts'][:n], z['eids'][:n], np.zeros(n, dtype=np.int64), seed=2, eval=True)
    res = []
    for on_device in (False, True):
        ds = mk()
        if on_device:
            ds.to_device(dev())
        for sampler in (None, ChunkSampler(n, rank=1, world_size=2, bs=B, seed=5)):
            model.reset()
            res.append(eval_utils.eval_edge_prediction(model, BatchLoader(ds, B, coll, sampler=sampler), dev(), restart_mode=False,
                                                       mean_over_n_samples=cfg['chunk']))
    assert res[0] == res[2] and res[1] == res[3] and res[0] != res[1]
