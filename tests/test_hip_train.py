"""GPU parity of the training tail (tg_train_step, tg_adam_step) against the oracle's autograd
gradients on the reference-generated training fixtures (tests/golden/train_*.npz)."""
import numpy as np
import pytest
import torch

from _util import load, parse_cfg, rel_err
from test_hip_parity import build_hip_model, dev
from test_oracle_golden import build_oracle, grad_err, TRAIN_FIXTURES, TRAIN_FIXTURES_L2

pytestmark = pytest.mark.gpu
TOL = 1e-4


def batch(z, cfg, b):
    B = cfg['B']
    lo, hi = b * B, min((b + 1) * B, len(z['src']))
    return [z[k][lo:hi] for k in ('src', 'dst', 'neg', 'ts', 'eids')]


def sync_params(model, orc):
    own = dict(model.named_parameters())
    with torch.no_grad():
        for k, v in orc.p.items():
            own[k].copy_(v.detach())


def lazy_restart(model, orc, cfg, b, a, cg, state):
    """train_self_supervised.py:152-163 on both sides"""
    if b == cfg.get('restart_at', -1):
        state['restarting'], state['uptodate'] = True, set()
        orc.clear_msgs()
        model.msg_store.clear()
    if state.get('restarting'):
        r_nodes = np.array(sorted(set(cg['involved'].tolist()) - state['uptodate']), dtype=np.int64)
        r_ts = np.full(len(r_nodes), np.float32(a[3]).min(), dtype=np.float32)
        orc.restart(r_nodes, r_ts)
        model.restart(torch.from_numpy(r_nodes), torch.from_numpy(r_ts))
        state['uptodate'].update(r_nodes.tolist())


@pytest.mark.parametrize('name', TRAIN_FIXTURES + TRAIN_FIXTURES_L2)
def test_contrast_gradients_match_oracle(name):
    from oracle import tiger_oracle as O
    from www2023tiger_amd.model.training import TrainBuffers
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg, dropout=0.0)
    orc = build_oracle(z, cfg)
    model.train()
    bufs = {}
    state = {}
    for b in range(cfg['n_batches']):
        a = batch(z, cfg, b)
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], cfg['K'], cfg['restarter'], hist_len=cfg.get('H'),
                       n_layers=cfg.get('L', 1))
        sync_params(model, orc)
        lazy_restart(model, orc, cfg, b, a, cg, state)
        c, _, grads = orc.train_step(*a, cg, lr=cfg['lr'], contrast_only=True)
        n = len(a[0])
        tb = bufs.get(n) or TrainBuffers(model, n)
        bufs[n] = tb
        to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()
        assert int(tb.sb.err.item()) == 0
        assert abs(float(tb.losses[0]) - c) < TOL * max(1.0, abs(c)), b
        for k, g in tb.grads.items():
            assert grad_err(g.cpu().numpy(), grads[k].numpy()) < 2e-4, (b, k)
        ukey = 'right_mem_updater.cell.bias_ih' if 'right_mem_updater.cell.bias_ih' in grads else 'right_mem_updater.fn.fc2.bias'
        gru_ran = float(grads[ukey].abs().max()) > 0
        assert int(tb.flags[0]) == 1 and int(tb.flags[1]) == int(gru_ran), b
    assert rel_err(model.left_memory.vals.cpu().numpy(), orc.left_vals.numpy()) < TOL
    assert rel_err(model.right_memory.vals.cpu().numpy(), orc.right_vals.numpy()) < TOL


def test_fused_trainer_follows_reference_trajectory():
    """tg_train_step + tg_adam_step, no parameter sync: the contrast-only trajectory of the
    reference (losses per batch, final parameters, final memories)."""
    from www2023tiger_amd.model.training import FusedTrainer
    z = load('train_contrast_rr_d8')
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg, dropout=0.0)
    model.train()
    tr = FusedTrainer(model, cfg['B'], lr=cfg['lr'])
    for b in range(cfg['n_batches']):
        losses = tr.step(*batch(z, cfg, b))
        assert abs(float(losses[0]) - float(z[f'b{b}_contrast_loss'])) < 1e-3, b
    for k, p in model.named_parameters():
        if k.startswith('restarter_fn.'):
            continue
        assert rel_err(p.detach().cpu().numpy(), z[f'final.{k}']) < 2e-3, k
    assert rel_err(model.left_memory.vals.cpu().numpy(), z['final_left_vals']) < 1e-3


def test_c2_shape_gradients_match_oracle():
    """d=172, B=1024, K=10 (BASELINE configs[1] shapes) after a few streaming batches: every
    gradient against the oracle's autograd."""
    import bench
    from oracle import tiger_oracle as O
    from www2023tiger_amd.model.training import TrainBuffers
    B = 1024
    st = bench.make_stream(2000, 300, 6 * B, 6.0e4, seed=3, d_e=172)
    model, orc = bench.build_models(st, 172, 10, 'left', 'left', with_oracle=True, dropout=0.0)
    model.train()
    tb = TrainBuffers(model, B)
    to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
    for b in range(5):
        a = [st[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], 10, 'static')
        if b < 4:  # stream forward on both sides
            orc.stream_step(*a, cg)
            model.stream_step(*a)
            continue
        c, _, grads = orc.train_step(*a, cg, lr=1e-3, contrast_only=True)
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()
        assert int(tb.sb.err.item()) == 0
        assert abs(float(tb.losses[0]) - c) < TOL * max(1.0, abs(c))
        worst = max((grad_err(g.cpu().numpy(), grads[k].numpy()), k) for k, g in tb.grads.items())
        assert worst[0] < 3e-4, worst


def test_c2_width_mutual_gradients_with_the_seq_restarter_match_oracle():
    """The training iteration of the reference's DEFAULT recipe at the benchmarked widths - d = 172 (d_model 860, two heads),
    hist_len 40, K = 10, all-zero node features (the narrow operand form) - on a batch of 192 after three streamed batches
    (its nodes have histories and memories): both losses and every gradient, the SeqRestarter's included, against the
    oracle's autograd (the fixtures cover d = 8 / 32)."""
    import bench
    from oracle import tiger_oracle as O
    from www2023tiger_amd.model.training import TrainBuffers
    B, d, K, H = 192, 172, 10, 40
    st = bench.make_stream(600, 120, 5 * B, 3.0e4, seed=11, d_e=d)
    model, orc = bench.build_models(st, d, K, 'left', 'right', restarter='seq', hist_len=H, with_oracle=True, dropout=0.0)
    model.train()
    assert model.restarter_fn.raw_feat_getter.nfeats_all_zero()
    tb = TrainBuffers(model, B, mutual=True)
    to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
    for b in range(4):
        a = [st[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'seq', hist_len=H)
        if b < 3:  # stream forward on both sides
            orc.stream_step(*a, cg)
            model.stream_step(*a)
            continue
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()  # (gradients only: the parameters are those the oracle differentiates at)
        c, ml, grads = orc.train_step(*a, cg, lr=1e-3, mutual_coef=1.0)
        assert int(tb.sb.err.item()) == 0
        assert abs(float(tb.losses[0]) - c) < TOL * max(1.0, abs(c))
        assert abs(float(tb.losses[1]) - ml) < TOL * max(1.0, abs(ml)), (float(tb.losses[1]), ml)
        worst = max((grad_err(g.cpu().numpy(), grads[k].numpy()), k) for k, g in tb.grads.items())
        assert worst[0] < 3e-4, worst
        assert any(k.startswith('restarter_fn.') and float(g.abs().max()) > 0 for k, g in tb.grads.items())


@pytest.mark.parametrize('strategy', ['recent_nodes', 'uniform'])
def test_training_step_with_other_sampling_strategies(strategy):
    """--strategy recent_nodes / uniform in the device training step (tg_step_io.strategy = 1 / 2): the neighbourhoods follow
    the graph's strategy (graph.py:94-148) while the hit windows of STEP 7 stay recent-edges lists (data_loader.py:61-66:
    one more sampler launch); loss and every gradient against the oracle's autograd over the same collation - with
    `uniform`, the same draws of the graph's MT19937 stream."""
    import bench
    from oracle import tiger_oracle as O
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.model.training import TrainBuffers
    B, K, d, nb = 256, 10, 32, 5
    st = bench.make_stream(300, 50, (nb + 1) * B, 2.0e4, seed=9, d_e=d)
    model, orc = bench.build_models(st, d, K, 'left', 'left', with_oracle=True, dropout=0.0)
    model.graph = Graph.from_arrays(st['src'], st['dst'], st['ts'], st['eids'], strategy=strategy, seed=4,
                                    max_node_id=st['n_nodes'] - 1, device=dev())
    orc.graph = O.OracleGraph(st['src'], st['dst'], st['ts'], st['eids'], strategy=strategy, seed=4, max_node_id=st['n_nodes'] - 1)
    model.train()
    tb = TrainBuffers(model, B)
    to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
    differs = False
    for b in range(nb):
        a = [st[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        sync_params(model, orc)
        c, _, grads = orc.train_step(*a, cg, lr=1e-3, contrast_only=True)
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()
        assert int(tb.sb.err.item()) == 0
        assert abs(float(tb.losses[0]) - c) < TOL * max(1.0, abs(c)), b
        worst = max((grad_err(g.cpu().numpy(), grads[k].numpy()), k) for k, g in tb.grads.items())
        assert worst[0] < 2e-4, (b, worst)
        edges = orc.graph.sample_temporal_neighbor(np.concatenate(a[:3]), np.tile(a[3], 3), K, strategy='recent_edges')[0]
        differs = differs or not np.array_equal(edges, cg['l1_nids'])
    assert differs  # the strategy really samples other lists than the hit windows use
    assert rel_err(model.left_memory.vals.cpu().numpy(), orc.left_vals.numpy()) < TOL
    assert rel_err(model.right_memory.vals.cpu().numpy(), orc.right_vals.numpy()) < TOL


@pytest.mark.parametrize('strategy', ['recent_nodes', 'uniform'])
def test_two_layer_training_step_with_the_recent_nodes_strategy(strategy):
    """--n_layers 2 --strategy recent_nodes / uniform in the device training step (init_utils.py:36-41 allows the
    combinations): both hops sampled with the graph's strategy (uniform: the same draws of the graph's MT19937 stream, first
    hop then second), the hit windows recent-edges lists; loss and the gradients of BOTH attention layers against the oracle's
    autograd over the same collation."""
    from oracle import tiger_oracle as O
    from www2023tiger_amd.model.training import TrainBuffers
    z = load('train_static_lr_d8_L2')
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg, strategy=strategy, dropout=0.0)
    orc = build_oracle(z, cfg)
    orc.graph = O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'], strategy=strategy, seed=0)
    model.train()
    bufs = {}
    for b in range(cfg['n_batches']):
        a = batch(z, cfg, b)
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], cfg['K'], cfg['restarter'], hist_len=cfg.get('H'), n_layers=2)
        sync_params(model, orc)
        c, _, grads = orc.train_step(*a, cg, lr=cfg['lr'], contrast_only=True)
        n = len(a[0])
        tb = bufs.get(n) or TrainBuffers(model, n)
        bufs[n] = tb
        to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()
        assert int(tb.sb.err.item()) == 0
        assert abs(float(tb.losses[0]) - c) < TOL * max(1.0, abs(c)), b
        # (uniform draws reach far back: time-encoding arguments of large time differences, float32 sums in another order -
        #  3.0e-4 of the largest entry on the second layer's merger, whose gradient entries are ~1e-5)
        for k, g in tb.grads.items():
            assert grad_err(g.cpu().numpy(), grads[k].numpy()) < (2e-4 if strategy == 'recent_nodes' else 4e-4), (b, k)
    assert rel_err(model.left_memory.vals.cpu().numpy(), orc.left_vals.numpy()) < TOL


@pytest.mark.parametrize('name', ['train_seq_lr_d8', 'train_static_ll_d16', 'train_mlp_merge_d8', 'train_linear_gru_d8',
                                  'train_static_lr_d8_L2', 'train_seq_lr_d8_zeronf'])
def test_mutual_gradients_match_oracle(name):
    """contrast + mutual loss (tiger.py:547-592): restarter gradients and both losses."""
    from oracle import tiger_oracle as O
    from www2023tiger_amd.model.training import TrainBuffers
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg, dropout=0.0)
    orc = build_oracle(z, cfg)
    model.train()
    bufs, state = {}, {}
    for b in range(cfg['n_batches']):
        a = batch(z, cfg, b)
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], cfg['K'], cfg['restarter'], hist_len=cfg.get('H'),
                       n_layers=cfg.get('L', 1))
        sync_params(model, orc)
        lazy_restart(model, orc, cfg, b, a, cg, state)
        rkey = next(k for k in orc.p if k.startswith('restarter_fn.') and 'time_encoder' not in k)
        before = orc.adam[rkey][2] if hasattr(orc, 'adam') else 0
        c, ml, grads = orc.train_step(*a, cg, lr=cfg['lr'], mutual_coef=1.0)
        had_grad = orc.adam[rkey][2] - before  # Adam counts a step only for parameters whose .grad is not None
        n = len(a[0])
        tb = bufs.get(n) or TrainBuffers(model, n, mutual=True)
        bufs[n] = tb
        to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()
        assert int(tb.sb.err.item()) == 0
        assert abs(float(tb.losses[0]) - c) < TOL * max(1.0, abs(c)), b
        assert abs(float(tb.losses[1]) - ml) < TOL * max(1.0, abs(ml)), (b, float(tb.losses[1]), ml)
        for k, g in tb.grads.items():
            assert grad_err(g.cpu().numpy(), grads[k].numpy()) < 2e-4, (b, k)
        assert int(tb.flags[2]) == had_grad, b


@pytest.mark.parametrize('name', ['train_seq_lr_d8', 'train_static_ll_d16'])
@pytest.mark.parametrize('loop', ['host_sets', 'device'])
def test_fused_trainer_mutual_trajectory(loop, name):
    """Full reference recipe (seq restarter, mutual learning, lazy restart at batch 6) with no
    parameter sync: per-batch losses and final parameters of the reference run.  host_sets: the loop's bookkeeping
    (train_self_supervised.py:152-163) as the reference writes it, with Python sets; device: FusedTrainer.enable_lazy_restart
    with the trigger pre-drawn - seq restarter: a collate-only pass lists `involved & ~uptodate` on the device, one count is
    read back; static restarter: the whole loop body inside the training step."""
    from www2023tiger_amd.model.training import FusedTrainer
    z = load(name)
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    model.train()
    tr = FusedTrainer(model, cfg['B'], lr=cfg['lr'], mutual=True, mutual_coef=cfg['mutual_coef'])
    restarting, uptodate = False, set()
    if loop == 'device':
        trigger = np.zeros(cfg['n_batches'], dtype=np.uint8)
        trigger[cfg['restart_at']] = 1
        tr.enable_lazy_restart(trigger)
    n_listed = 0
    for b in range(cfg['n_batches']):
        a = batch(z, cfg, b)
        if loop == 'device':
            losses = tr.step(*a)
            n_listed += tr.restarted if cfg['restarter'] == 'seq' else int(tr.buf.sb.counts[3])
            assert n_listed == 0 or b >= cfg['restart_at']
            assert abs(float(losses[0]) - float(z[f'b{b}_contrast_loss'])) < 1e-3, b
            assert abs(float(losses[1]) - float(z[f'b{b}_mutual_loss'])) < 1e-3, b
            continue
        if b == cfg['restart_at']:
            restarting, uptodate = True, set()
            model.msg_store.clear()
        if restarting:
            cg = coll.collate_arrays(*a)[-1]
            r_nodes = np.array(sorted(set(cg.np_computation_graph_nodes.tolist()) - uptodate), dtype=np.int64)
            model.restart(torch.from_numpy(r_nodes), torch.full((len(r_nodes),), float(np.float32(a[3]).min())))
            uptodate.update(r_nodes.tolist())
        losses = tr.step(*a)
        assert abs(float(losses[0]) - float(z[f'b{b}_contrast_loss'])) < 1e-3, b
        assert abs(float(losses[1]) - float(z[f'b{b}_mutual_loss'])) < 1e-3, b
    assert loop != 'device' or n_listed > 0
    for k, p in model.named_parameters():
        assert rel_err(p.detach().cpu().numpy(), z[f'final.{k}']) < 2e-3, k


@pytest.mark.parametrize('name,strategy', [('train_seq_lr_d8', 'recent_nodes'), ('train_static_lr_d8_L2', 'recent_edges'),
                                           ('train_static_lr_d8_L2', 'recent_nodes')])
def test_fused_trainer_device_loop_equals_host_sets_on_other_forms(name, strategy):
    """FusedTrainer.enable_lazy_restart beside the loop written with Python sets, where no reference trajectory exists: a
    `recent_nodes` graph (the list form's collate-only pass must sample with the graph's strategy) and two layers (both hops
    are involved: in-step loop of the static restarter) - same losses, same parameters after every iteration."""
    from www2023tiger_amd.model.training import FusedTrainer
    z = load(name)
    cfg = parse_cfg(z)
    out = {}
    for loop in ('host_sets', 'device'):
        model, _, coll = build_hip_model(z, cfg, strategy=strategy, dropout=0.0)
        model.train()
        tr = FusedTrainer(model, cfg['B'], lr=cfg['lr'], mutual=True, mutual_coef=cfg['mutual_coef'])
        restarting, uptodate, losses = False, set(), []
        if loop == 'device':
            trigger = np.zeros(cfg['n_batches'], dtype=np.uint8)
            trigger[cfg['restart_at']] = 1
            tr.enable_lazy_restart(trigger)
        for b in range(cfg['n_batches']):
            a = batch(z, cfg, b)
            if loop == 'host_sets':
                if b == cfg['restart_at']:
                    restarting, uptodate = True, set()
                    model.msg_store.clear()
                if restarting:
                    cg = coll.collate_arrays(*a)[-1]
                    r_nodes = np.array(sorted(set(cg.np_computation_graph_nodes.tolist()) - uptodate), dtype=np.int64)
                    model.restart(torch.from_numpy(r_nodes), torch.full((len(r_nodes),), float(np.float32(a[3]).min())))
                    uptodate.update(r_nodes.tolist())
            losses.append(tr.step(*a).clone())
        out[loop] = (losses, {k: p.detach().clone() for k, p in model.named_parameters()}, model.left_memory.vals.clone())
    # (two runs of the same iteration agree to rounding only: the backward pass accumulates with atomics, Adam at lr 1e-2
    #  carries the last bits on)
    for la, lb in zip(out['host_sets'][0], out['device'][0]):
        assert torch.allclose(la, lb, rtol=1e-4, atol=1e-5)
    for k, p in out['host_sets'][1].items():
        assert rel_err(out['device'][1][k].cpu().numpy(), p.cpu().numpy()) < TOL, k
    assert rel_err(out['device'][2].cpu().numpy(), out['host_sets'][2].cpu().numpy()) < TOL


def test_restart_list_in_train_mode_draws_the_masks_of_restart():
    """TIGER.restart_list with the SeqRestarter in train() mode (the training script's lazy-restart loop calls restart with
    dropout active): one library call (tg_restart_seq_list_train) - the memories TIGER.restart leaves from the same generator
    state, bit for bit, the generator advanced alike; and not the rows of the inference form."""
    z = load('train_seq_lr_d8')
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg, dropout=0.1)
    model.train()
    nids = torch.arange(1, int(z['n_nodes']), device=dev())
    t = float(np.float32(z['ts'].max())) * 0.7
    tdev = torch.tensor([t], dtype=torch.float32, device=dev())
    rng = model.dropout_rng()
    start = rng.clone()
    out = {}
    for form in ('restart', 'list', 'list_eval'):
        model.reset()
        rng.copy_(start)
        if form == 'list_eval':
            model.eval()
        if form == 'restart':
            model.restart(nids, tdev.expand(len(nids)))
        else:
            model.restart_list(nids, tdev)
        out[form] = (model.left_memory.vals.clone(), model.right_memory.vals.clone(), model.left_memory.update_ts.clone(), rng.clone())
    for a_, b_ in zip(out['restart'], out['list']):
        assert torch.equal(a_, b_)
    assert not torch.equal(out['list'][3], start)                 # the generator moved ...
    assert not torch.equal(out['list'][1], out['list_eval'][1])   # ... and the masks were applied


def test_tables_derived_from_parameters_follow_the_device_optimizer():
    """tg_adam_step updates parameters through raw pointers - torch's version counters, the stamps of everything derived
    from parameters (the SeqRestarter's tabulated anony_emb block, pre-multiplied attention weights, eager-update and
    per-node tables), do not move.  train() starts a new parameter epoch instead: after inference -> training with the
    device optimizer -> inference on the SAME model, the restarter's rows and a streamed batch's embeddings equal those of a
    fresh model that carries the trained parameters."""
    from www2023tiger_amd.model.training import FusedTrainer
    z = load('train_seq_lr_d8_zeronf')
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg, dropout=0.0)
    model.eval()
    model.fuse_attention()
    model.eager_updates()
    nids = torch.arange(1, int(z['n_nodes']), device=dev())
    ts = torch.full((len(nids),), float(np.float32(z['ts'].max())), device=dev())
    with torch.no_grad():
        before = model.restarter_fn(nids, ts)[0].clone()  # (fills the table of the anony_emb block)
        model.stream_step(*batch(z, cfg, 0), lean=True)    # (builds the eager-update / per-node tables)
    assert model.restarter_fn._ta_cache is not None
    model.train()
    model.reset()
    tr = FusedTrainer(model, cfg['B'], lr=5e-2, mutual=True)
    for b in range(4):
        tr.step(*batch(z, cfg, b))
    fresh, _, _ = build_hip_model(z, cfg, dropout=0.0)
    fresh.load_state_dict({k: v.detach().clone() for k, v in model.state_dict().items()})
    model.eval(), fresh.eval()
    model.reset(), fresh.reset()
    with torch.no_grad():
        own, ref = model.restarter_fn(nids, ts), fresh.restarter_fn(nids, ts)
        assert float((own[0] - before).abs().max()) > 1e-3  # the parameters did move
        for a_, b_ in zip(own, ref):
            assert torch.equal(a_, b_)
        for b in range(3):
            h_own = model.stream_step(*batch(z, cfg, b), lean=True).h.clone()
            h_ref = fresh.stream_step(*batch(z, cfg, b)).h.clone()
            assert rel_err(h_own.cpu().numpy(), h_ref.cpu().numpy()) < TOL, b


@pytest.mark.parametrize('which', ['torch', 'device-flags'])
def test_reference_training_loop_with_torch_adam(which):
    """The loop of train_self_supervised.py:143-171 written against the mirrored API only
    (collator -> contrast_and_mutual_learning -> loss.backward() -> Adam.step()): losses and final
    parameters of the reference run, with torch.optim.Adam (idle groups handed over as None, one host
    read-back per backward) and with www2023tiger_amd.optim.Adam (live flags stay on the device)."""
    from www2023tiger_amd import optim as tg_optim
    z = load('train_seq_lr_d8')
    cfg = parse_cfg(z)
    model, _, coll = build_hip_model(z, cfg, dropout=0.0)
    optimizer = (torch.optim.Adam if which == 'torch' else tg_optim.Adam)(model.parameters(), lr=cfg['lr'])
    model.train()
    model.reset()
    restarting, uptodate = False, set()
    device = dev()
    for b in range(cfg['n_batches']):
        src_ids, dst_ids, neg_dst_ids, ts, eids, _, comp_graph = coll.collate_arrays(*batch(z, cfg, b))
        src_ids, dst_ids, neg_dst_ids = (x.long().to(device) for x in (src_ids, dst_ids, neg_dst_ids))
        ts, eids = ts.float().to(device), eids.long().to(device)
        comp_graph.to(device)
        optimizer.zero_grad()
        if b == cfg['restart_at']:
            restarting, uptodate = True, set()
            model.msg_store.clear()
        if restarting:
            restart_nodes = set(comp_graph.np_computation_graph_nodes) - set(uptodate)
            r_nids = torch.tensor(sorted(restart_nodes)).long().to(device)
            model.restart(r_nids, torch.full((len(r_nids),), ts.min().item()).to(device))
            uptodate.update(restart_nodes)
        contrast_loss, mutual_loss = model.contrast_and_mutual_learning(
            src_ids, dst_ids, neg_dst_ids, ts, eids, comp_graph, contrast_only=False)
        loss = contrast_loss + cfg['mutual_coef'] * mutual_loss
        loss.backward()
        optimizer.step()
        assert abs(contrast_loss.item() - float(z[f'b{b}_contrast_loss'])) < 1e-3, b
        assert abs(mutual_loss.item() - float(z[f'b{b}_mutual_loss'])) < 1e-3, b
        if b in cfg['grad_batches'] and b < 2:  # identical parameters so far: gradients comparable directly
            for k, p in model.named_parameters():
                ref = z[f'b{b}_grad.{k}']
                got = np.zeros_like(ref) if p.grad is None else p.grad.cpu().numpy()
                assert grad_err(got, ref) < 2e-4, (b, k)
    for k, p in model.named_parameters():
        assert rel_err(p.detach().cpu().numpy(), z[f'final.{k}']) < 2e-3, k
    if which != 'torch':  # the deferred path really ran: gradients are views of the step's flat buffer
        tb = model._step_ws[('train', len(src_ids), True)]
        assert all(p.grad is tb.grads[k] for k, p in model.named_parameters() if k in tb.grads)
        model.flush_msg()  # polls the deferred invariant word
    # evaluation mode still takes the inference path
    model.eval()
    model.reset()
    with torch.no_grad():
        out = model.contrast_and_mutual_learning(src_ids, dst_ids, neg_dst_ids, ts, eids, comp_graph)
    assert out[0].grad_fn is None


def test_dropout_mask_generator_matches_host_mirror():
    """keep rate and the exact mask: run a dropout-only probe (score-head hidden layer) is
    indirect, so compare through full parity below; here the host mirror's statistics."""
    from oracle.tiger_oracle import dropout_keep
    k = dropout_keep(123, 5, 2, 200000, 0.1)
    assert abs(k.mean() - 0.9) < 3e-3
    assert (dropout_keep(123, 5, 2, 1000, 0.1) != dropout_keep(123, 6, 2, 1000, 0.1)).any()
    assert (dropout_keep(123, 5, 2, 1000, 0.1) != dropout_keep(123, 5, 3, 1000, 0.1)).any()
    np.testing.assert_array_equal(dropout_keep(123, 5, 2, 10, 0.1, offset=7), dropout_keep(123, 5, 2, 17, 0.1)[7:])


@pytest.mark.parametrize('name', ['train_seq_lr_d8', 'train_static_ll_d16'])
def test_training_with_dropout_matches_oracle_given_the_same_masks(name):
    """dropout 0.1 / 0.3 at all four sites: losses and every gradient against the oracle fed with the
    host mirror of the library's masks (same seed and step counter)."""
    from oracle import tiger_oracle as O
    from www2023tiger_amd.model.training import TrainBuffers
    z = load(name)
    cfg = parse_cfg(z)
    p = 0.1 if 'seq' in name else 0.3
    model, _, _ = build_hip_model(z, cfg, dropout=p)
    orc = build_oracle(z, cfg)
    model.train()
    bufs, state = {}, {}
    seed = 987654321
    model.dropout_rng().copy_(torch.tensor([seed, 0], dtype=torch.int64))
    orc.dropout = (p, seed, 0)
    for b in range(cfg['n_batches']):
        a = batch(z, cfg, b)
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], cfg['K'], cfg['restarter'], hist_len=cfg.get('H'))
        sync_params(model, orc)
        lazy_restart(model, orc, cfg, b, a, cg, state)  # train() mode: the restarter's dropout is active here too
        assert int(model.dropout_rng()[1]) == orc.dropout[2], b
        n = len(a[0])
        tb = bufs.get(n)
        if tb is None:
            tb = bufs[n] = TrainBuffers(model, n, mutual=True)
        c, ml, grads = orc.train_step(*a, cg, lr=cfg['lr'], mutual_coef=1.0)
        to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()
        assert int(tb.rng[1]) == orc.dropout[2]
        assert abs(float(tb.losses[0]) - c) < TOL * max(1.0, abs(c)), b
        assert abs(float(tb.losses[1]) - ml) < TOL * max(1.0, abs(ml)), b
        for k, g in tb.grads.items():
            assert grad_err(g.cpu().numpy(), grads[k].numpy()) < 2e-4, (b, k)
    # evaluation ignores dropout
    model.eval()
    assert rel_err(model.left_memory.vals.cpu().numpy(), orc.left_vals.numpy()) < TOL


@pytest.mark.parametrize('d,n_head,hit,K,restarter', [(32, 4, 'count', 6, 'seq'), (16, 1, 'vec', 6, 'static'),
                                                     (256, 2, 'bin', 10, 'seq')])
def test_training_other_shapes(d, n_head, hit, K, restarter):
    """heads 1 / 4, d = 256 (two float4 per lane and segment), 'count' / 'vec' hit features: forward,
    every gradient and both losses against the oracle on a synthetic stream."""
    import bench
    from oracle import tiger_oracle as O
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.model.feature_getter import NumericalFeature
    from www2023tiger_amd.model.restarters import SeqRestarter, StaticRestarter
    from www2023tiger_amd.model.tiger import TIGER
    from www2023tiger_amd.model.training import TrainBuffers
    B, H = 64, 8
    st = bench.make_stream(60, 20, 6 * B, 400.0, seed=11, d_e=d)
    n_nodes = st['n_nodes']
    g = Graph.from_arrays(st['src'], st['dst'], st['ts'], st['eids'], strategy='recent_edges', seed=0,
                          max_node_id=n_nodes - 1, device=dev())
    rs = np.random.RandomState(3)
    nfeats = (rs.standard_normal((n_nodes, d)) * 0.3).astype(np.float32)
    nfeats[0] = 0
    torch.manual_seed(5)
    fg = NumericalFeature(torch.from_numpy(nfeats), torch.from_numpy(st['efeats']), dim=d, device=dev())
    fg.n_nodes, fg.n_edges = n_nodes, len(st['src'])
    rst = (SeqRestarter(raw_feat_getter=fg, graph=g, hist_len=H, n_head=n_head, dropout=0.0) if restarter == 'seq'
           else StaticRestarter(raw_feat_getter=fg, graph=g))
    model = TIGER(raw_feat_getter=fg, graph=g, restarter=rst, n_neighbors=K, hit_type=hit, n_layers=1, n_head=n_head,
                  dropout=0.0, msg_src='left', upd_src='right').to(dev())
    with torch.no_grad():
        model.time_encoder.phase.uniform_(-0.5, 0.5)
        if restarter == 'static':
            rst.left_emb.weight.normal_(0, 0.1)
            rst.right_emb.weight.normal_(0, 0.1)
    og = O.OracleGraph(st['src'], st['dst'], st['ts'], st['eids'], max_node_id=n_nodes - 1)
    params = {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}
    orc = O.OracleTIGER(params, og, n_nodes=n_nodes, dim=d, nfeats=nfeats, efeats=st['efeats'], n_neighbors=K,
                        msg_src='left', upd_src='right', restarter=restarter, hist_len=H, n_head=n_head, hit_type=hit)
    model.train()
    tb = TrainBuffers(model, B, mutual=True)
    to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
    for b in range(5):
        a = [st[k][b * B:(b + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, restarter, hist_len=H)
        sync_params(model, orc)
        c, ml, grads = orc.train_step(*a, cg, lr=1e-3, mutual_coef=1.0)
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()
        assert int(tb.sb.err.item()) == 0
        assert abs(float(tb.losses[0]) - c) < TOL * max(1.0, abs(c)), b
        assert abs(float(tb.losses[1]) - ml) < TOL * max(1.0, abs(ml)), b
        worst = max((grad_err(g_.cpu().numpy(), grads[k].numpy()), k) for k, g_ in tb.grads.items())
        assert worst[0] < 3e-4, (b, worst)


@pytest.mark.parametrize('name,H', [('train_seq_lr_d8_zeronf', 100), ('train_seq_lr_d8', 72), ('train_seq_lr_d8_zeronf', 33)])
def test_long_history_restarter_forward_and_gradients(name, H):
    """--hist_len beyond 64 (init_utils.py:58 takes any int; the score kernels' 128-row grids need more dynamic LDS than a
    launch gets by default) and just above 32 (both row classes of the 64-row launch pair): the SeqRestarter's forward on
    sampled histories and the mutual-learning gradients against the oracle - on a zero node-feature table (narrow form of
    the Q / K projection) and on a random one (wide form).  The fixtures' streams with parameters regenerated for the
    longer history (the anony_emb table grows with it)."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from _weights import golden_param
    from oracle import tiger_oracle as O
    from _util import fixture_tables
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.model.feature_getter import NumericalFeature
    from www2023tiger_amd.model.restarters import SeqRestarter
    from www2023tiger_amd.model.tiger import TIGER
    from www2023tiger_amd.model.training import TrainBuffers
    z = load(name)
    cfg = parse_cfg(z)
    cfg['H'] = H
    n_nodes, nfeats, efeats = fixture_tables(z, cfg)
    g = Graph.from_arrays(z['src'], z['dst'], z['ts'], z['eids'], strategy='recent_edges', seed=0, device=dev())
    fg = NumericalFeature(None if nfeats is None else torch.from_numpy(nfeats), torch.from_numpy(efeats), dim=cfg['d'], device=dev())
    fg.n_nodes, fg.n_edges = n_nodes, len(z['src'])
    rst = SeqRestarter(raw_feat_getter=fg, graph=g, hist_len=H, n_head=2, dropout=0.0)
    model = TIGER(raw_feat_getter=fg, graph=g, restarter=rst, n_neighbors=cfg['K'], hit_type=cfg.get('hit', 'bin'), n_layers=1,
                  n_head=2, dropout=0.0, msg_src=cfg['msg_src'], upd_src=cfg['upd_src'])
    params = {k: golden_param(k, tuple(v.shape), cfg['wseed']) for k, v in model.named_parameters()}
    with torch.no_grad():
        for k, v in model.named_parameters():
            v.copy_(torch.from_numpy(params[k]))
    model = model.to(dev())
    og = O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'], strategy='recent_edges', seed=0)
    orc = O.OracleTIGER(params, og, n_nodes=n_nodes, dim=cfg['d'], nfeats=nfeats, efeats=efeats, n_neighbors=cfg['K'],
                        msg_src=cfg['msg_src'], upd_src=cfg['upd_src'], restarter='seq', hist_len=H, hit_type=cfg.get('hit', 'bin'))
    # forward at the end of the stream: long histories for the popular nodes, short (padded) ones for the others
    model.eval()
    nids = np.arange(1, n_nodes, dtype=np.int64)
    ts = np.full(len(nids), np.float32(z['ts'].max()) + 1.0, dtype=np.float32)
    hl, hr, pt = model.restarter_fn(torch.from_numpy(nids).to(dev()), torch.from_numpy(ts).to(dev()))
    rl, rr, rp = orc.restarter_forward(nids, ts)
    assert rel_err(hl.cpu().numpy(), rl.detach().numpy()) < TOL and rel_err(hr.cpu().numpy(), rr.detach().numpy()) < TOL
    np.testing.assert_array_equal(pt.cpu().numpy(), rp.numpy())
    # gradients through the mutual loss
    model.train()
    bufs = {}
    for b in range(4):
        a = batch(z, cfg, b + 5)  # later batches: histories have filled up
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], cfg['K'], 'seq', hist_len=H)
        sync_params(model, orc)
        c, ml, grads = orc.train_step(*a, cg, lr=cfg['lr'], mutual_coef=1.0)
        n = len(a[0])
        tb = bufs.get(n) or TrainBuffers(model, n, mutual=True)
        bufs[n] = tb
        to = lambda x, dt: torch.as_tensor(x).to(dev(), dt)
        tb.sb.load(to(a[0], torch.int64), to(a[1], torch.int64), to(a[2], torch.int64), to(a[3], torch.float64),
                   to(a[4], torch.int64))
        tb.launch()
        assert int(tb.sb.err.item()) == 0
        assert abs(float(tb.losses[1]) - ml) < TOL * max(1.0, abs(ml)), (b, float(tb.losses[1]), ml)
        for k, gv in tb.grads.items():
            assert grad_err(gv.cpu().numpy(), grads[k].numpy()) < 2e-4, (b, k)


@pytest.mark.parametrize('name,H', [('train_seq_lr_d8_zeronf', 1), ('train_seq_lr_d8_zeronf', 2), ('train_seq_lr_d8', 1), ('train_seq_lr_d8', 5)])
def test_restarter_forward_on_empty_and_minimal_histories(name, H):
    """Edge cases of the compact-row restarter (csrc/tg_restart.hip): hist_len 1 (only the shared last row exists), 2 and 5;
    every node restarted at time 0 (no history at all: every slot padded, the last one included), at a mid-stream time and
    at the end; a single-node call; both operand forms - against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from _weights import golden_param
    from oracle import tiger_oracle as O
    from _util import fixture_tables
    from www2023tiger_amd.data.graph import Graph
    from www2023tiger_amd.model.feature_getter import NumericalFeature
    from www2023tiger_amd.model.restarters import SeqRestarter
    from www2023tiger_amd.model.tiger import TIGER
    z = load(name)
    cfg = parse_cfg(z)
    n_nodes, nfeats, efeats = fixture_tables(z, cfg)
    g = Graph.from_arrays(z['src'], z['dst'], z['ts'], z['eids'], strategy='recent_edges', seed=0, device=dev())
    fg = NumericalFeature(None if nfeats is None else torch.from_numpy(nfeats), torch.from_numpy(efeats), dim=cfg['d'], device=dev())
    fg.n_nodes, fg.n_edges = n_nodes, len(z['src'])
    rst = SeqRestarter(raw_feat_getter=fg, graph=g, hist_len=H, n_head=2, dropout=0.0)
    model = TIGER(raw_feat_getter=fg, graph=g, restarter=rst, n_neighbors=cfg['K'], hit_type=cfg.get('hit', 'bin'), n_layers=1,
                  n_head=2, dropout=0.0, msg_src=cfg['msg_src'], upd_src=cfg['upd_src'])
    params = {k: golden_param(k, tuple(v.shape), cfg['wseed']) for k, v in model.named_parameters()}
    with torch.no_grad():
        for k, v in model.named_parameters():
            v.copy_(torch.from_numpy(params[k]))
    model = model.to(dev()).eval()
    og = O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'], strategy='recent_edges', seed=0)
    orc = O.OracleTIGER(params, og, n_nodes=n_nodes, dim=cfg['d'], nfeats=nfeats, efeats=efeats, n_neighbors=cfg['K'],
                        msg_src=cfg['msg_src'], upd_src=cfg['upd_src'], restarter='seq', hist_len=H, hit_type=cfg.get('hit', 'bin'))
    tmax = float(np.float32(z['ts'].max()))
    for nids, t in ((np.arange(n_nodes, dtype=np.int64), 0.0), (np.arange(1, n_nodes, dtype=np.int64), 0.5 * tmax),
                    (np.arange(1, n_nodes, dtype=np.int64), tmax + 1.0), (np.array([int(z['dst'][0])], dtype=np.int64), tmax + 1.0)):
        ts = np.full(len(nids), np.float32(t), dtype=np.float32)
        with torch.no_grad():
            hl, hr, pt = model.restarter_fn(torch.from_numpy(nids).to(dev()), torch.from_numpy(ts).to(dev()))
        rl, rr, rp = orc.restarter_forward(nids, ts)
        assert rel_err(hl.cpu().numpy(), rl.detach().numpy()) < TOL, (H, t)
        assert rel_err(hr.cpu().numpy(), rr.detach().numpy()) < TOL, (H, t)
        np.testing.assert_array_equal(pt.cpu().numpy(), rp.numpy())
        # the list form (one library call, device-resident time) leaves the same memories as restart()
        ref_model_left = model.left_memory.vals.clone()
        model.restart(torch.from_numpy(nids).to(dev()), torch.from_numpy(ts).to(dev()))
        a = (model.left_memory.vals.clone(), model.right_memory.vals.clone(), model.left_memory.update_ts.clone())
        model.left_memory.vals.copy_(ref_model_left)
        model.restart_list(torch.from_numpy(nids).to(dev()), torch.tensor([t], dtype=torch.float32, device=dev()))
        assert torch.equal(a[0], model.left_memory.vals) and torch.equal(a[1], model.right_memory.vals)
        assert torch.equal(a[2], model.left_memory.update_ts)
    # several lists in one forward, each at its own time (tg_restart_seq_lists_fwd; an empty list among them): the rows of the
    # per-list calls, bit for bit
    tdev = lambda t: torch.tensor([t], dtype=torch.float32, device=dev())
    ids_all = torch.arange(1, n_nodes, dtype=torch.int64, device=dev())
    cut = (n_nodes - 1) // 3
    lists = [ids_all[:cut], ids_all[:0], ids_all[cut:cut + 1], ids_all[cut + 1:]]
    times = [tdev(0.3 * tmax), tdev(0.0), tdev(0.0), tdev(tmax + 1.0)]
    ids, hl, hr, pt = model.restart_lists_forward(lists, times)
    assert torch.equal(ids, ids_all)
    at = 0
    for nl, tl in zip(lists, times):
        k = int(nl.numel())
        if not k:
            continue
        one = [torch.empty(k, cfg['d'], device=dev()), torch.empty(k, cfg['d'], device=dev()), torch.empty(k, device=dev())]
        model.restart_list_forward(nl, tl, *one)
        for a_, b_ in zip(one, (hl[at:at + k], hr[at:at + k], pt[at:at + k])):
            assert torch.equal(a_, b_)
        at += k
