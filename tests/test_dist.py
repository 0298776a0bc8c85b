"""Multi-GPU host logic (www2023tiger_amd/dist.py): shard plan, row mapping, the
all-gather exchange and the replicated write-back.

* CPU, world_size 2, gloo: the runner drives an oracle-backed compute object; the result
  must equal the single-process oracle on the same global batches (bit for bit - the same
  float32 ops run in both).
* GPU (one card, two ranks sharing it, gloo staging): the HIP backend under the same
  runner must reproduce the single-GPU engine on the global batch.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from _util import fixture_params, fixture_tables, load, parse_cfg, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_owner_table_and_plan():
    from www2023tiger_amd.dist import ShardPlan, balanced_owner_table
    rs = np.random.RandomState(0)
    n_nodes, world, cap = 200, 4, 64
    p = 1.0 / np.arange(1, 101)
    dst_all = rs.choice(100, 5000, p=p / p.sum()) + 100
    owner = balanced_owner_table(n_nodes, dst_all, world)
    assert owner.min() >= 0 and owner.max() < world
    load = np.bincount(owner[dst_all], minlength=world)
    deg = np.bincount(dst_all, minlength=n_nodes)
    assert load.max() <= max(deg.max(), 1.05 * load.mean() + deg.max() * 0.0 + 1) or load.max() <= 1.1 * load.mean()
    dst = dst_all[:128]
    plan = ShardPlan(dst, owner, world, cap)
    assert plan.counts.sum() == 128
    seen = np.concatenate(plan.local_idx)
    assert sorted(seen.tolist()) == list(range(128))          # every event embedded exactly once
    for r in range(world):
        assert (owner[dst[plan.local_idx[r]]] == r).all()       # on the owner of its dst
        assert (np.diff(plan.local_idx[r]) > 0).all()           # stream order kept inside a shard
    assert len(set(plan.left_row.tolist())) == 256 and len(set(plan.new_row.tolist())) == 256
    assert not set(plan.left_row.tolist()) & set(plan.new_row.tolist())
    with pytest.raises(ValueError):
        ShardPlan(dst, owner, world, 8)
    # capacity-balanced plan: exactly Bg/world events per rank, owner placement kept while there is room
    bal = ShardPlan(dst, owner, world, 32, balance=True, layout=(5 * 32, 0, 3 * 32, 32))
    assert bal.counts.tolist() == [32] * world
    moved = bal.rank_of != owner[dst]
    for r in range(world):  # an event only leaves its owner when the owner's shard is full
        if moved[owner[dst] == r].any():
            assert (bal.rank_of == r).sum() == 32
    assert sorted(np.concatenate(bal.local_idx).tolist()) == list(range(128))
    rows = np.concatenate([bal.left_row, bal.new_row])
    assert len(set(rows.tolist())) == 512 and rows.max() < world * 5 * 32
    assert ((bal.left_row % (5 * 32)) < 2 * 32).all() and ((bal.new_row % (5 * 32)) >= 3 * 32).all()


def test_plan_row_maps_of_the_bench_layout_at_eight_ranks():
    """The layout ResidentShardedStream gathers (4B rows per rank: [h(t'+) src | h(t'+) dst | h(t-) src | h(t-) dst])
    at the size the driver runs: 8 ranks x B = 1024, Zipf destinations, every shard exactly full."""
    from www2023tiger_amd.dist import ShardPlan, balanced_owner_table
    rs = np.random.RandomState(3)
    world, B, n_items = 8, 1024, 1000
    p = 1.0 / np.arange(1, n_items + 1)
    dst_all = rs.choice(n_items, 40000, p=p / p.sum()) + 8228
    owner = balanced_owner_table(9228, dst_all, world)
    for b in range(4):
        dst = dst_all[b * world * B:(b + 1) * world * B]
        plan = ShardPlan(dst, owner, world, B, balance=True, layout=(4 * B, 2 * B, 0, B))
        assert plan.counts.tolist() == [B] * world
        assert sorted(np.concatenate(plan.local_idx).tolist()) == list(range(world * B))
        rows = np.concatenate([plan.left_row, plan.new_row])
        assert len(set(rows.tolist())) == 4 * world * B and rows.min() >= 0 and rows.max() < world * 4 * B
        assert ((plan.left_row % (4 * B)) >= 2 * B).all() and ((plan.new_row % (4 * B)) < 2 * B).all()
        # position i of cat[src, dst] of the global batch: rank of its event, role block (src / dst), slot in the shard
        ev = np.tile(np.arange(world * B), 2)
        role = np.repeat([0, 1], world * B)
        assert (plan.left_row // (4 * B) == plan.rank_of[ev]).all()
        assert (((plan.left_row % (4 * B)) - 2 * B) // B == role).all() and ((plan.new_row % (4 * B)) // B == role).all()


# ------------------------------------------------------------------------------ oracle backend (CPU)
def test_hash_owner_table_needs_no_stream_and_spreads_the_ids():
    """the literal "destination-node hash": a function of (node id, world) only; ids spread evenly, a plan over it shards a
    batch by owner(dst) like the balanced table does"""
    from www2023tiger_amd.dist import ShardPlan, hash_owner_table
    for world in (2, 3, 8):
        o = hash_owner_table(9228, world)
        assert o.min() == 0 and o.max() == world - 1
        cnt = np.bincount(o, minlength=world)
        assert cnt.max() - cnt.min() <= 0.02 * cnt.mean() + 8
        np.testing.assert_array_equal(o, hash_owner_table(9228, world))  # deterministic
        np.testing.assert_array_equal(o[:5000], hash_owner_table(5000, world))  # a node's owner does not depend on n_nodes
    rs = np.random.RandomState(0)
    dst = rs.randint(8228, 9228, 256)
    o = hash_owner_table(9228, 4)
    p = ShardPlan(dst, o, 4, 256)
    np.testing.assert_array_equal(p.rank_of, o[dst])


class OracleBackend:
    """Splits OracleTIGER.contrast_learning (tiger.py:196-255) into the two halves the
    runner needs.  Test-only: uses oracle/ as the compute."""

    def __init__(self, orc, K, restarter):
        from oracle import tiger_oracle as O
        self.O, self.orc, self.K, self.restarter = O, orc, K, restarter

    def embed(self, src, dst, neg, ts, eids):
        O, m = self.O, self.orc
        if len(src) == 0:
            z = torch.zeros(0, m.d)
            return z, z
        cg = O.collate(m.graph, src, dst, neg, ts, self.K, 'static')
        t32 = torch.from_numpy(np.asarray(ts)).float()
        involved = cg['involved']
        outdated, h_new, _ = m.consume(involved)
        reprs = m.right_vals[torch.from_numpy(involved)].clone()
        if len(outdated):
            reprs[torch.from_numpy(cg['local_index'][outdated])] = h_new
        nids3 = np.concatenate([src, dst, neg])
        h = m.embed(reprs, cg['local_index'], nids3, t32.repeat(3), cg['l1_nids'], cg['l1_eids'], cg['l1_ts'])
        pos = np.concatenate([src, dst])
        return h[:2 * len(src)], reprs[torch.from_numpy(cg['local_index'][pos])]

    def writeback(self, src, dst, ts, eids, rows, left_row, new_row):
        O, m = self.O, self.orc
        t32 = torch.from_numpy(np.asarray(ts)).float()
        pos = np.concatenate([src, dst])
        ts2 = t32.repeat(2)
        ids, idx = O.select_latest_nids(pos, ts2.numpy())
        had = m.has_msg[ids]
        if had.any():  # STEP 4
            sel = ids[had]
            m.has_msg[sel] = False
            m._mem_set(m.right_vals, m.right_ts, torch.from_numpy(sel), rows[torch.from_numpy(new_row[idx[had]])],
                       m.msg_ts[torch.from_numpy(sel)])
        m.store_events(src, dst, t32, eids)  # STEP 5
        m._mem_set(m.left_vals, m.left_ts, torch.from_numpy(ids), rows[torch.from_numpy(left_row[idx])],
                   ts2[torch.from_numpy(idx)])  # STEP 6


    # ---- PeriodicShardedRunner: state rows of a node list as one [n, W] block (left | ts | right | ts | mailbox | ts)
    device = torch.device('cpu')

    def row_width(self):
        m = self.orc
        return 2 * (m.d + 1) + m.msg_vals.shape[1] + 1

    def export_rows(self, ids):
        m, i = self.orc, torch.from_numpy(np.asarray(ids, dtype=np.int64))
        return torch.cat([m.left_vals[i], m.left_ts[i, None], m.right_vals[i], m.right_ts[i, None], m.msg_vals[i],
                          m.msg_ts[i, None]], 1)

    def import_rows(self, ids, rows):
        m, ids = self.orc, np.asarray(ids, dtype=np.int64)
        i, d, mw = torch.from_numpy(ids), self.orc.d, self.orc.msg_vals.shape[1]
        m.left_vals[i], m.left_ts[i] = rows[:, :d], rows[:, d]
        m.right_vals[i], m.right_ts[i] = rows[:, d + 1:2 * d + 1], rows[:, 2 * d + 1]
        m.msg_vals[i], m.msg_ts[i] = rows[:, 2 * d + 2:2 * d + 2 + mw], rows[:, 2 * d + 2 + mw]
        m.has_msg[ids] = True


def _make_oracle(z, cfg):
    from oracle import tiger_oracle as O
    n_nodes, nfeats, efeats = fixture_tables(z, cfg)
    g = O.OracleGraph(z['src'], z['dst'], z['ts'], z['eids'])
    return O.OracleTIGER(fixture_params(z, cfg), g, n_nodes=n_nodes, dim=cfg['d'], nfeats=nfeats, efeats=efeats,
                         n_neighbors=cfg['K'], msg_src=cfg['msg_src'], upd_src=cfg['upd_src'], restarter='static')


def _cpu_worker(rank, world, port, name, Bg, n_batches, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from www2023tiger_amd.dist import ShardedRunner, balanced_owner_table
    z = load(name)
    cfg = parse_cfg(z)
    orc = _make_oracle(z, cfg)
    owner = balanced_owner_table(int(z['n_nodes']), z['dst'], world)
    runner = ShardedRunner(OracleBackend(orc, cfg['K'], 'static'), owner, rank, world, cap=Bg)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        runner.step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), left=orc.left_vals.numpy(), right=orc.right_vals.numpy(),
             left_ts=orc.left_ts.numpy(), right_ts=orc.right_ts.numpy(), msg=orc.msg_vals.numpy(),
             msg_ts=orc.msg_ts.numpy(), has=orc.has_msg)
    tdist.destroy_process_group()


def test_sharded_equals_single_process_cpu_gloo(tmp_path):
    name, Bg, n_batches, world = 'static_ll_d16', 96, 6, 2
    mp.spawn(_cpu_worker, args=(world, free_port(), name, Bg, n_batches, str(tmp_path)), nprocs=world, join=True)
    from oracle import tiger_oracle as O
    z = load(name)
    cfg = parse_cfg(z)
    ref = _make_oracle(z, cfg)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        ref.contrast_learning(*a, O.collate(ref.graph, a[0], a[1], a[2], a[3], cfg['K'], 'static'))
    want = dict(left=ref.left_vals.numpy(), right=ref.right_vals.numpy(), left_ts=ref.left_ts.numpy(),
                right_ts=ref.right_ts.numpy(), msg_ts=ref.msg_ts.numpy(), has=ref.has_msg)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))
        for k, v in want.items():
            np.testing.assert_array_equal(got[k], v, err_msg=f'rank {r} {k}')
        np.testing.assert_array_equal(got['msg'][ref.has_msg], ref.msg_vals.numpy()[ref.has_msg])


def _cpu_period_worker(rank, world, port, name, Bg, n_batches, period, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from www2023tiger_amd.dist import PeriodicShardedRunner, balanced_owner_table
    z = load(name)
    cfg = parse_cfg(z)
    orc = _make_oracle(z, cfg)
    owner = balanced_owner_table(int(z['n_nodes']), z['dst'], world)
    runner = PeriodicShardedRunner(OracleBackend(orc, cfg['K'], 'static'), owner, rank, world, cap=Bg, period=period)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        runner.step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    runner.flush()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), left=orc.left_vals.numpy(), right=orc.right_vals.numpy(),
             left_ts=orc.left_ts.numpy(), right_ts=orc.right_ts.numpy(), msg=orc.msg_vals.numpy(),
             msg_ts=orc.msg_ts.numpy(), has=orc.has_msg, sent=np.int64(runner.exchanged_rows))
    tdist.destroy_process_group()


def _single_process_state(name, Bg, n_batches):
    from oracle import tiger_oracle as O
    z = load(name)
    cfg = parse_cfg(z)
    ref = _make_oracle(z, cfg)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        ref.contrast_learning(*a, O.collate(ref.graph, a[0], a[1], a[2], a[3], cfg['K'], 'static'))
    return ref


@pytest.mark.parametrize('world', [2, 3])
def test_exchange_period_one_equals_single_process_cpu_gloo(tmp_path, world):
    """period 1 of the periodic exchange (write back the own shard, all-gather the touched rows, latest event wins) is the
    exact global write-back: every replica equals the single-process engine bit for bit."""
    name, Bg, n_batches = 'static_ll_d16', 96, 6
    mp.spawn(_cpu_period_worker, args=(world, free_port(), name, Bg, n_batches, 1, str(tmp_path)), nprocs=world, join=True)
    ref = _single_process_state(name, Bg, n_batches)
    want = dict(left=ref.left_vals.numpy(), right=ref.right_vals.numpy(), left_ts=ref.left_ts.numpy(),
                right_ts=ref.right_ts.numpy(), msg_ts=ref.msg_ts.numpy(), has=ref.has_msg)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))
        for k, v in want.items():
            np.testing.assert_array_equal(got[k], v, err_msg=f'rank {r} {k}')
        np.testing.assert_array_equal(got['msg'][ref.has_msg], ref.msg_vals.numpy()[ref.has_msg])


@pytest.mark.parametrize('period', [2, 4])
def test_exchange_period_k_replicas_agree_and_drift_is_bounded_cpu_gloo(tmp_path, period):
    """period k > 1: after the closing exchange all replicas hold the same state; times, the has-message set and the
    bookkeeping equal the exact engine's (they do not depend on stale rows), the float rows drift from it by a bounded
    amount (neighbour rows up to k - 1 batches stale) - and really do drift, i.e. the period is in effect."""
    name, Bg, n_batches, world = 'static_ll_d16', 96, 8, 2
    mp.spawn(_cpu_period_worker, args=(world, free_port(), name, Bg, n_batches, period, str(tmp_path)), nprocs=world,
             join=True)
    got = [np.load(os.path.join(str(tmp_path), f'rank{r}.npz')) for r in range(world)]
    for k in ('left', 'right', 'left_ts', 'right_ts', 'msg', 'msg_ts', 'has'):
        np.testing.assert_array_equal(got[0][k], got[1][k], err_msg=k)
    ref = _single_process_state(name, Bg, n_batches)
    np.testing.assert_array_equal(got[0]['left_ts'], ref.left_ts.numpy())
    np.testing.assert_array_equal(got[0]['has'], ref.has_msg)
    np.testing.assert_array_equal(got[0]['msg_ts'], ref.msg_ts.numpy())
    drift = np.abs(got[0]['left'] - ref.left_vals.numpy()).max() / max(1.0, np.abs(ref.left_vals.numpy()).max())
    assert 0.0 < drift < 0.5, drift


# ------------------------------------------------------------------------------ HIP backend (GPU)
def _gpu_worker(rank, world, port, name, Bg, n_batches, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    from test_hip_parity import build_hip_model
    from www2023tiger_amd.dist import HipBackend, ShardedRunner, balanced_owner_table
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    owner = balanced_owner_table(int(z['n_nodes']), z['dst'], world)
    backend = HipBackend(model, cap=Bg)
    runner = ShardedRunner(backend, owner, rank, world, cap=Bg)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        runner.step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    backend.check_invariants()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), left=model.left_memory.vals.cpu().numpy(),
             right=model.right_memory.vals.cpu().numpy(), left_ts=model.left_memory.update_ts.cpu().numpy(),
             right_ts=model.right_memory.update_ts.cpu().numpy(), msg=model.msg_store.node_msg_vals.cpu().numpy(),
             msg_ts=model.msg_store.node_msg_ts.cpu().numpy(),
             has=np.array(sorted(model.msg_store.nodes_with_messages), dtype=np.int64))
    tdist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['static_ll_d16', 'seq_rr_d8_nofeat'])
def test_sharded_equals_single_gpu(tmp_path, name):
    from test_hip_parity import build_hip_model
    Bg, n_batches, world = 100, 4, 2
    mp.spawn(_gpu_worker, args=(world, free_port(), name, Bg, n_batches, str(tmp_path)), nprocs=world, join=True)
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        model.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    has = np.array(sorted(model.msg_store.nodes_with_messages), dtype=np.int64)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))
        np.testing.assert_array_equal(got['left_ts'], model.left_memory.update_ts.cpu().numpy())
        np.testing.assert_array_equal(got['right_ts'], model.right_memory.update_ts.cpu().numpy())
        np.testing.assert_array_equal(got['has'], has)
        np.testing.assert_array_equal(got['msg_ts'], model.msg_store.node_msg_ts.cpu().numpy())
        assert rel_err(got['left'], model.left_memory.vals.cpu().numpy()) < 1e-6
        assert rel_err(got['right'], model.right_memory.vals.cpu().numpy()) < 1e-6
        assert rel_err(got['msg'][has], model.msg_store.node_msg_vals.cpu().numpy()[has]) < 1e-6


def _gpu_period_worker(rank, world, port, name, Bg, n_batches, period, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    from test_hip_parity import build_hip_model
    from www2023tiger_amd.dist import HipBackend, PeriodicShardedRunner, balanced_owner_table
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    owner = balanced_owner_table(int(z['n_nodes']), z['dst'], world)
    backend = HipBackend(model, cap=Bg)
    runner = PeriodicShardedRunner(backend, owner, rank, world, cap=Bg, period=period)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        runner.step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    runner.flush()
    backend.check_invariants()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), left=model.left_memory.vals.cpu().numpy(),
             right=model.right_memory.vals.cpu().numpy(), left_ts=model.left_memory.update_ts.cpu().numpy(),
             right_ts=model.right_memory.update_ts.cpu().numpy(), msg=model.msg_store.node_msg_vals.cpu().numpy(),
             msg_ts=model.msg_store.node_msg_ts.cpu().numpy(),
             has=np.array(sorted(model.msg_store.nodes_with_messages), dtype=np.int64))
    tdist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('period', [1, 3])
def test_exchange_period_on_the_hip_engine(tmp_path, period):
    """PeriodicShardedRunner over the HIP backend (two processes sharing the test box's GPU, gloo): period 1 equals the
    single-GPU engine (times and sets exactly, float rows to 1e-6: the shard's products are the global batch's on fewer rows);
    period 3: the replicas agree with each other exactly after the closing exchange, times / has-message set equal the exact
    engine's."""
    from test_hip_parity import build_hip_model
    name, Bg, n_batches, world = 'static_ll_d16', 100, 6, 2
    mp.spawn(_gpu_period_worker, args=(world, free_port(), name, Bg, n_batches, period, str(tmp_path)), nprocs=world, join=True)
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        model.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    has = np.array(sorted(model.msg_store.nodes_with_messages), dtype=np.int64)
    got = [np.load(os.path.join(str(tmp_path), f'rank{r}.npz')) for r in range(world)]
    for k in ('left', 'right', 'left_ts', 'right_ts', 'msg_ts', 'has'):
        np.testing.assert_array_equal(got[0][k], got[1][k], err_msg=k)
    np.testing.assert_array_equal(got[0]['msg'][has], got[1]['msg'][has])
    np.testing.assert_array_equal(got[0]['left_ts'], model.left_memory.update_ts.cpu().numpy())
    np.testing.assert_array_equal(got[0]['has'], has)
    np.testing.assert_array_equal(got[0]['msg_ts'], model.msg_store.node_msg_ts.cpu().numpy())
    if period == 1:
        assert rel_err(got[0]['left'], model.left_memory.vals.cpu().numpy()) < 1e-6
        assert rel_err(got[0]['right'], model.right_memory.vals.cpu().numpy()) < 1e-6
        assert rel_err(got[0]['msg'][has], model.msg_store.node_msg_vals.cpu().numpy()[has]) < 1e-6


def _gpu_resident_worker(rank, world, port, name, B, n_steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    from test_hip_parity import build_hip_model
    from www2023tiger_amd.dist import ResidentShardedStream, balanced_owner_table
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    stream = {k: z[k] for k in ('src', 'dst', 'neg', 'ts', 'eids')}
    owner = balanced_owner_table(int(z['n_nodes']), z['dst'], world)
    rs = ResidentShardedStream(model, stream, owner, rank, world, B, n_steps)
    rs.step()                     # eager
    torch.cuda.synchronize()
    rs.capture()                  # remaining steps replay the two captured hipGraphs
    for _ in range(n_steps - 1):
        rs.step()
    rs.check_invariants()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), left=model.left_memory.vals.cpu().numpy(),
             right=model.right_memory.vals.cpu().numpy(), left_ts=model.left_memory.update_ts.cpu().numpy(),
             right_ts=model.right_memory.update_ts.cpu().numpy(), msg=model.msg_store.node_msg_vals.cpu().numpy(),
             msg_ts=model.msg_store.node_msg_ts.cpu().numpy(),
             has=np.array(sorted(model.msg_store.nodes_with_messages), dtype=np.int64))
    tdist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('world,B', [(2, 48), (4, 24)])
def test_resident_sharded_stream_graph_replay_equals_single_gpu(tmp_path, world, B):
    """The production multi-GPU loop (balanced shards, resident stream, two hipGraphs + one
    all-gather per step) against the single-GPU fused step on the same global batches; two and four
    ranks (processes sharing the one GPU of the test box, gloo for the exchange)."""
    from test_hip_parity import build_hip_model
    name, n_steps = 'static_ll_d16', 6
    mp.spawn(_gpu_resident_worker, args=(world, free_port(), name, B, n_steps, str(tmp_path)), nprocs=world, join=True)
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    Bg = B * world
    for b in range(n_steps):
        sl = slice(b * Bg, (b + 1) * Bg)
        model.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    has = np.array(sorted(model.msg_store.nodes_with_messages), dtype=np.int64)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f'rank{r}.npz'))
        np.testing.assert_array_equal(got['left_ts'], model.left_memory.update_ts.cpu().numpy())
        np.testing.assert_array_equal(got['right_ts'], model.right_memory.update_ts.cpu().numpy())
        np.testing.assert_array_equal(got['has'], has)
        assert rel_err(got['left'], model.left_memory.vals.cpu().numpy()) < 1e-6
        assert rel_err(got['right'], model.right_memory.vals.cpu().numpy()) < 1e-6
        assert rel_err(got['msg'][has], model.msg_store.node_msg_vals.cpu().numpy()[has]) < 1e-6


# ------------------------------------------------------------------------------ partitioned node state
class OraclePartitionEngine:
    """The local compute of www2023tiger_amd.dist.PartitionedRunner on the CPU oracle (test-only): a rank's
    OracleTIGER holds full-height tables of which only the rows of its own nodes are authoritative."""

    def __init__(self, orc, K):
        from oracle import tiger_oracle as O
        self.O, self.orc, self.K = O, orc, K
        self.device, self.d = torch.device('cpu'), orc.d

    def select_latest(self, pos, ts32):
        u, i = self.O.select_latest_nids(pos.numpy(), ts32.numpy())
        return torch.from_numpy(u), torch.from_numpy(i)

    def collate(self, src, dst, neg, ts):
        cg = self.O.collate(self.orc.graph, src.numpy(), dst.numpy(), neg.numpy(), ts.numpy(), self.K, 'static')
        return torch.from_numpy(cg['involved'])

    def _eff(self, ids):
        """right memory as STEP 4 would leave it: updater row / message time for nodes with a pending message"""
        m, ids_np = self.orc, ids.numpy()
        rows, ts = m.right_vals[ids].clone(), m.right_ts[ids].clone()
        has = m.has_msg[ids_np]
        if has.any():
            outd, h_new, mts = m.consume(ids_np[has])
            where = torch.from_numpy(np.searchsorted(outd, ids_np[has]))
            sel = torch.from_numpy(np.nonzero(has)[0])
            rows[sel], ts[sel] = h_new[where], mts[where]
        return rows, ts

    def serve(self, p):
        m = self.orc
        out = torch.empty(len(p.serve_eff) + len(p.serve_msg), self.d + 1)
        r, t = self._eff(p.serve_eff)
        r2, t2 = (m.left_vals[p.serve_msg], m.left_ts[p.serve_msg]) if m.msg_src == 'left' else self._eff(p.serve_msg)
        out[p.serve_eff_pos], out[p.serve_msg_pos] = torch.cat([r, t[:, None]], 1), torch.cat([r2, t2[:, None]], 1)
        return out

    def adopt(self, p, got):
        m, d = self.orc, self.d
        eff_rows, msg_rows = got[p.reply_eff_pos], got[p.reply_msg_pos]
        assert not m.has_msg[p.req_eff.numpy()].any()  # a rank never holds a message of a node it does not own
        m.right_vals[p.req_eff], m.right_ts[p.req_eff] = eff_rows[:, :d], eff_rows[:, d]
        vals, tss = m._mem(m.msg_src)
        vals[p.req_msg], tss[p.req_msg] = msg_rows[:, :d], msg_rows[:, d]

    def embed(self, p):
        O, m = self.O, self.orc
        rows = torch.zeros(3 * p.n + p.n_recv, self.d)
        if p.n == 0:
            return rows
        src, dst, neg, ts, _ = (x.numpy() for x in p.local)
        cg = O.collate(m.graph, src, dst, neg, ts, self.K, 'static')
        involved = cg['involved']
        outdated, h_new, _ = m.consume(involved)
        reprs = m.right_vals[torch.from_numpy(involved)].clone()
        if len(outdated):
            reprs[torch.from_numpy(cg['local_index'][outdated])] = h_new
        nids3 = np.concatenate([src, dst, neg])
        rows[:3 * p.n] = m.embed(reprs, cg['local_index'], nids3, p.local[3].float().repeat(3), cg['l1_nids'], cg['l1_eids'],
                                 cg['l1_ts'])
        return rows

    def writeback(self, p, rows, owner, rank):
        O, m = self.O, self.orc
        src, dst, ts, eids = p.glob
        Bg = len(src)
        t32 = ts.float()
        ts2 = t32.repeat(2)
        ids, idx = O.select_latest_nids(torch.cat([src, dst]).numpy(), ts2.numpy())
        keep = owner.numpy()[ids] == rank
        ids, idx = ids[keep], idx[keep]
        np.testing.assert_array_equal(ids, p.mine.numpy())          # the plan's winners are the write-back's
        np.testing.assert_array_equal(idx, p.mine_index.numpy())
        had = m.has_msg[ids]
        if had.any():  # STEP 4 (tiger.py:230-241)
            sel = ids[had]
            outd, h_new, mts = m.consume(sel)
            np.testing.assert_array_equal(outd, sel)
            m.has_msg[sel] = False
            m._mem_set(m.right_vals, m.right_ts, torch.from_numpy(sel), h_new, mts)
        own, e = torch.from_numpy(ids), torch.from_numpy(idx % Bg)  # STEP 5 (tiger.py:422-442) for this rank's nodes
        other = torch.where(torch.from_numpy(idx) < Bg, dst[e], src[e])
        mv, mt = m._mem(m.msg_src)
        te, opt_ = t32[e], mt[own]
        if (opt_ > te).any():
            raise ValueError('Events occur before the udpated memory.')
        msg = torch.cat([mv[own] + m.node_feat(own), mv[other] + m.node_feat(other), m.edge_feat(eids[e]),
                         m.te(te - opt_)], 1)
        m.msg_vals[own], m.msg_ts[own] = msg, te
        m.has_msg[ids] = True
        m._mem_set(m.left_vals, m.left_ts, own, rows[p.left_row[torch.from_numpy(idx)]], ts2[torch.from_numpy(idx)])  # STEP 6

    def refresh(self, p):
        pass  # the oracle computes updater rows on demand


def _save_owned(path, owner, rank, left, right, left_ts, right_ts, msg, msg_ts, has):
    np.savez(path, owner=owner, rank=rank, left=left, right=right, left_ts=left_ts, right_ts=right_ts, msg=msg,
             msg_ts=msg_ts, has=has)


def _assemble_owned(out_dir, world):
    """the authoritative rows of every rank put back into full tables"""
    parts = [np.load(os.path.join(out_dir, f'rank{r}.npz')) for r in range(world)]
    owner = parts[0]['owner']
    full = {}
    for k in ('left', 'right', 'left_ts', 'right_ts', 'msg', 'msg_ts', 'has'):
        a = np.zeros_like(parts[0][k])
        for r, p in enumerate(parts):
            np.testing.assert_array_equal(p['owner'], owner)
            a[owner == r] = p[k][owner == r]
        full[k] = a
    return full, owner


def _partitioned_cpu_worker(rank, world, port, name, Bg, n_batches, balance, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from www2023tiger_amd.dist import PartitionedRunner, ShardPlan, balanced_owner_table
    z = load(name)
    cfg = parse_cfg(z)
    orc = _make_oracle(z, cfg)
    owner = balanced_owner_table(int(z['n_nodes']), z['dst'], world)
    runner = PartitionedRunner(OraclePartitionEngine(orc, cfg['K']), owner, rank, world)
    pulled = pushed = 0
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        rank_of = ShardPlan(z['dst'][sl], owner, world, Bg // world, balance=True).rank_of if balance else None
        plan = runner.plan(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')), rank_of=rank_of)
        runner.run(plan)
        pulled += plan.stats['pulled_rows']
        pushed += plan.stats['pushed_rows']
    assert pulled > 0 and pushed > 0   # the exchange really carried rows
    _save_owned(os.path.join(out_dir, f'rank{rank}.npz'), owner, rank, orc.left_vals.numpy(), orc.right_vals.numpy(),
                orc.left_ts.numpy(), orc.right_ts.numpy(), orc.msg_vals.numpy(), orc.msg_ts.numpy(), orc.has_msg)
    tdist.destroy_process_group()


@pytest.mark.parametrize('world,balance', [(2, False), (2, True), (4, True)])
def test_partitioned_state_equals_single_process_cpu_gloo(tmp_path, world, balance):
    """Node state partitioned by owner(node), one all_to_all of rows each way per global batch (pull, push):
    the owners' rows after 6 global batches are those of the single-process oracle on the same batches, bit for
    bit; with owner placement of the events and with capacity-balanced placement (an event may then run on a rank
    that owns neither endpoint)."""
    name, Bg, n_batches = 'static_ll_d16', 96, 6
    mp.spawn(_partitioned_cpu_worker, args=(world, free_port(), name, Bg, n_batches, balance, str(tmp_path)),
             nprocs=world, join=True)
    from oracle import tiger_oracle as O
    z = load(name)
    cfg = parse_cfg(z)
    ref = _make_oracle(z, cfg)
    torch.set_num_threads(1)
    for b in range(n_batches):
        sl = slice(b * Bg, (b + 1) * Bg)
        a = [z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        ref.contrast_learning(*a, O.collate(ref.graph, a[0], a[1], a[2], a[3], cfg['K'], 'static'))
    got, owner = _assemble_owned(str(tmp_path), world)
    assert len(set(owner.tolist())) == world
    np.testing.assert_array_equal(got['has'], ref.has_msg)
    np.testing.assert_array_equal(got['left_ts'], ref.left_ts.numpy())
    np.testing.assert_array_equal(got['right_ts'], ref.right_ts.numpy())
    np.testing.assert_array_equal(got['msg_ts'][ref.has_msg], ref.msg_ts.numpy()[ref.has_msg])
    # two ranks: bit for bit.  Four ranks embed 24 events each: the CPU BLAS rounds a product of 72 rows differently
    # from the same rows inside a product of 288 (4e-8 absolute), which is all that separates the float tables
    same = np.testing.assert_array_equal if world == 2 else (lambda a, b, err_msg='': np.testing.assert_allclose(
        a, b, rtol=0, atol=2e-7, err_msg=err_msg))
    same(got['left'], ref.left_vals.numpy(), err_msg='left')
    same(got['right'], ref.right_vals.numpy(), err_msg='right')
    same(got['msg'][ref.has_msg], ref.msg_vals.numpy()[ref.has_msg], err_msg='mailbox')


def _partitioned_gpu_worker(rank, world, port, name, B, n_steps, resident, out_dir, lopsided=False, physical=False,
                            exchange='rccl', graph_steps=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    from test_hip_parity import build_hip_model
    from www2023tiger_amd.dist import (HipPartitionEngine, PartitionedRunner, ResidentPartitionedStream,
                                       balanced_owner_table)
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    model.fuse_attention()
    Bg = B * world
    owner = balanced_owner_table(int(z['n_nodes']), z['dst'], world)
    if lopsided:  # rank 0 owns every node: the other ranks embed their share of the events and own no winner, ever
        owner = np.zeros_like(owner)
    keys = ('src', 'dst', 'neg', 'ts', 'eids')
    if resident:  # all plans up front, balanced shards (the benchmarked form)
        rs = ResidentPartitionedStream(model, {k: z[k] for k in keys}, owner, rank, world, B, n_steps, physical=physical,
                                       exchange=exchange)
        eng = rs.engine
        if graph_steps:  # one eager step, then graphs of `graph_steps` steps (the window form: tg_part_step)
            rs.step()
            tdist.barrier()
            rs.capture_steps(graph_steps)
            tdist.barrier()
            while rs.steps_done + graph_steps <= n_steps:
                rs.replay()
            torch.cuda.synchronize()
            while rs.steps_done < n_steps:
                rs.step()
        else:
            for _ in range(n_steps):
                rs.step()
        torch.cuda.synchronize()
        rs.check_invariants()
        tdist.barrier()  # nobody unmaps a window a peer may still be storing into
    else:         # plan + run per batch, events on the owner of their destination
        eng = HipPartitionEngine(model, cap=Bg)
        if physical:
            eng.partition(owner, rank, arena_rows=int(z['n_nodes']))
        runner = PartitionedRunner(eng, owner, rank, world)
        for b in range(n_steps):
            runner.step(*(z[k][b * Bg:(b + 1) * Bg] for k in keys))
        eng.check_invariants()
    if physical:  # this rank's tables hold its own rows (+ an arena): scattered back to node ids for the comparison
        assert model.left_memory.vals.shape[0] == 1 + eng.n_own + max(eng.arena_rows, 1)  # row 0 | own rows | arena
        assert world == 1 or eng.n_own < int(z['n_nodes']) - 1                              # (the ranks really share the nodes)
        f = eng.export_full()
        _save_owned(os.path.join(out_dir, f'rank{rank}.npz'), owner, rank, f['left'], f['right'], f['left_ts'], f['right_ts'],
                    f['msg'], f['msg_ts'], f['has'])
    else:
        has = model.msg_store.has_msg_mask().cpu().numpy()
        _save_owned(os.path.join(out_dir, f'rank{rank}.npz'), owner, rank, model.left_memory.vals.cpu().numpy(),
                    model.right_memory.vals.cpu().numpy(), model.left_memory.update_ts.cpu().numpy(),
                    model.right_memory.update_ts.cpu().numpy(), model.msg_store.node_msg_vals.cpu().numpy(),
                    model.msg_store.node_msg_ts.cpu().numpy(), has)
    tdist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('name,world,B,resident,physical', [
    ('static_ll_d16', 2, 48, False, False), ('static_ll_d16', 4, 24, True, False), ('seq_rr_d8_nofeat', 2, 50, True, False),
    ('static_ll_d16', 2, 48, False, True), ('static_ll_d16', 4, 24, True, True), ('seq_rr_d8_nofeat', 2, 50, True, True)])
def test_partitioned_state_equals_single_gpu(tmp_path, name, world, B, resident, physical):
    """The partitioned mode on the HIP engine (ranks are processes sharing the one GPU of the test box, gloo for
    the exchange) against the single-GPU fused step on the same global batches: the owners' rows.  physical: every
    rank's state tables hold only row 0, its own nodes' rows and an arena for the rows pulled per batch
    (tg_model.row_of) - fewer rows than nodes - and state is addressed by row."""
    from test_hip_parity import build_hip_model
    n_steps = min(6, len(load(name)['src']) // (B * world))
    mp.spawn(_partitioned_gpu_worker, args=(world, free_port(), name, B, n_steps, resident, str(tmp_path), False, physical),
             nprocs=world, join=True)
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    model.fuse_attention()
    model.eager_updates()
    Bg = B * world
    for b in range(n_steps):
        sl = slice(b * Bg, (b + 1) * Bg)
        model.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    got, _ = _assemble_owned(str(tmp_path), world)
    has = model.msg_store.has_msg_mask().cpu().numpy()
    np.testing.assert_array_equal(got['has'], has)
    np.testing.assert_array_equal(got['left_ts'], model.left_memory.update_ts.cpu().numpy())
    np.testing.assert_array_equal(got['right_ts'], model.right_memory.update_ts.cpu().numpy())
    np.testing.assert_array_equal(got['msg_ts'][has], model.msg_store.node_msg_ts.cpu().numpy()[has])
    assert rel_err(got['left'], model.left_memory.vals.cpu().numpy()) < 1e-6
    assert rel_err(got['right'], model.right_memory.vals.cpu().numpy()) < 1e-6
    assert rel_err(got['msg'][has], model.msg_store.node_msg_vals.cpu().numpy()[has]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('name,world,B,physical,graph_steps', [
    ('static_ll_d16', 2, 48, True, 0), ('static_ll_d16', 4, 24, True, 0), ('seq_rr_d8_nofeat', 2, 50, False, 0),
    ('static_ll_d16', 2, 48, True, 2), ('static_ll_d16', 4, 24, False, 2), ('static_ll_d16', 1, 96, True, 2)])
def test_partitioned_window_exchange_equals_single_gpu(tmp_path, name, world, B, physical, graph_steps):
    """The partitioned step as ONE library call per global batch (tg_part_step): the two exchanges are kernels that store
    into the peers' exported windows (hipIpcGetMemHandle; the ranks are processes sharing the test box's one GPU) and
    signal with epoch flags - no collective in the step.  Eager calls and replays of a captured graph of two steps, two
    and four ranks (and the one-rank point of the form), physical partition and full-height tables: the owners' rows
    against the single-GPU fused step on the same global batches."""
    from test_hip_parity import build_hip_model
    n_steps = min(6, len(load(name)['src']) // (B * world))
    mp.spawn(_partitioned_gpu_worker, args=(world, free_port(), name, B, n_steps, True, str(tmp_path), False, physical, 'ipc',
                                            graph_steps), nprocs=world, join=True)
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    model.fuse_attention()
    model.eager_updates()
    Bg = B * world
    for b in range(n_steps):
        sl = slice(b * Bg, (b + 1) * Bg)
        model.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    got, _ = _assemble_owned(str(tmp_path), world)
    has = model.msg_store.has_msg_mask().cpu().numpy()
    np.testing.assert_array_equal(got['has'], has)
    np.testing.assert_array_equal(got['left_ts'], model.left_memory.update_ts.cpu().numpy())
    np.testing.assert_array_equal(got['right_ts'], model.right_memory.update_ts.cpu().numpy())
    np.testing.assert_array_equal(got['msg_ts'][has], model.msg_store.node_msg_ts.cpu().numpy()[has])
    assert rel_err(got['left'], model.left_memory.vals.cpu().numpy()) < 1e-6
    assert rel_err(got['right'], model.right_memory.vals.cpu().numpy()) < 1e-6
    assert rel_err(got['msg'][has], model.msg_store.node_msg_vals.cpu().numpy()[has]) < 1e-6


@pytest.mark.gpu
def test_partitioned_rank_that_owns_no_node(tmp_path):
    """A rank whose write-back is empty in every batch (it owns no node; its events' rows are all pushed away) must
    neither fail nor stall its peers inside the step's collectives (an empty winner list has a NULL data
    pointer, which the write-back took for the unplanned form)."""
    from test_hip_parity import build_hip_model
    name, world, B = 'static_ll_d16', 2, 48
    n_steps = min(5, len(load(name)['src']) // (B * world))
    mp.spawn(_partitioned_gpu_worker, args=(world, free_port(), name, B, n_steps, True, str(tmp_path), True),
             nprocs=world, join=True)
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg)
    model.fuse_attention()
    model.eager_updates()
    Bg = B * world
    for b in range(n_steps):
        sl = slice(b * Bg, (b + 1) * Bg)
        model.stream_step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    r0 = np.load(os.path.join(str(tmp_path), 'rank0.npz'))  # rank 0 is authoritative for everything
    has = model.msg_store.has_msg_mask().cpu().numpy()
    np.testing.assert_array_equal(r0['has'], has)
    np.testing.assert_array_equal(r0['left_ts'], model.left_memory.update_ts.cpu().numpy())
    assert rel_err(r0['left'], model.left_memory.vals.cpu().numpy()) < 1e-6
    assert rel_err(r0['right'], model.right_memory.vals.cpu().numpy()) < 1e-6
    assert rel_err(r0['msg'][has], model.msg_store.node_msg_vals.cpu().numpy()[has]) < 1e-6


# ------------------------------------------------------------------------------ the reference's DDP recipe
def _ddp_worker(rank, world, port, root, n_steps, out_dir):
    """train_self_supervised_ddp.py:107-211 on this package's API: ChunkSampler time chunks,
    always-on lazy restarts, torch DDP (find_unused_parameters) around the model, Adam with lr*sqrt(world)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    from torch.nn.parallel import DistributedDataParallel as DDP
    from www2023tiger_amd.init_utils import init_data, init_model
    device = torch.device('cuda', 0)
    torch.manual_seed(0)
    basic, (train_graph, full_graph), dls = init_data(
        'toy', root, 0, rank=rank, world_size=world, num_workers=0, bs=64, warmup_steps=0, subset=1.0,
        strategy='recent_edges', n_layers=1, n_neighbors=4, restarter_type='seq', hist_len=6, device=device)
    model = init_model(basic[0], basic[1], train_graph, full_graph, basic[2], device, dim=8, n_layers=1, n_heads=2,
                       n_neighbors=4, hit_type='bin', dropout=0.0, restarter_type='seq', hist_len=6, msg_src='left',
                       upd_src='right', msg_tsfm_type='id', mem_update_type='gru')
    ddp_model = DDP(model, broadcast_buffers=False, find_unused_parameters=True)
    optimizer = torch.optim.Adam(ddp_model.parameters(), lr=1e-3 * np.sqrt(world))
    ddp_model.train()
    model.reset()
    uptodate, losses = set(), []
    first_index = None
    for i_batch, (src, dst, neg, ts, eids, _, cg) in enumerate(dls[0]):
        if i_batch == n_steps:
            break
        if first_index is None:
            first_index = int(eids[0])
        src, dst, neg, eids = (x.long().to(device) for x in (src, dst, neg, eids))
        ts = ts.float().to(device)
        optimizer.zero_grad()
        todo = set(cg.np_computation_graph_nodes.tolist()) - uptodate  # DDP mode: always restarting
        nids = torch.tensor(sorted(todo), dtype=torch.long, device=device)
        model.restart(nids, torch.full((len(nids),), ts.min().item(), device=device))
        uptodate |= todo
        c_loss, m_loss = ddp_model(src, dst, neg, ts, eids, cg, contrast_only=False)
        loss = c_loss + m_loss
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu().numpy()
    np.savez(os.path.join(out_dir, f'ddp{rank}.npz'), params=flat, losses=np.array(losses), first=first_index)
    tdist.destroy_process_group()


@pytest.mark.gpu
def test_reference_ddp_recipe_two_ranks(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_input_side import write_files
    z = load('input_side')
    write_files(str(tmp_path), 'toy', z, with_feats=False)
    world, n_steps = 2, 4
    mp.spawn(_ddp_worker, args=(world, free_port(), str(tmp_path), n_steps, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(os.path.join(str(tmp_path), f'ddp{r}.npz')) for r in range(world))
    assert np.isfinite(r0['losses']).all() and np.isfinite(r1['losses']).all()
    assert int(r0['first']) != int(r1['first'])                    # different time chunks
    assert not np.allclose(r0['losses'], r1['losses'])               # ... hence different losses
    np.testing.assert_array_equal(r0['params'], r1['params'])        # gradients were all-reduced: replicas agree


def _fused_ddp_worker(rank, world, port, n_steps, out_dir):
    """FusedTrainer with one all-reduce of the flat gradient buffer per iteration: every rank trains its own
    time chunk (ChunkSampler ranges), replicas stay identical; rank 0 also dumps its per-step gradients."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    from test_hip_parity import build_hip_model
    from www2023tiger_amd.model.training import FusedTrainer
    z = load('train_seq_lr_d8')
    cfg = parse_cfg(z)
    model, _, _ = build_hip_model(z, cfg, dropout=0.0)
    model.train()
    tr = FusedTrainer(model, cfg['B'], lr=cfg['lr'], mutual=True, world_size=world)
    B = cfg['B']
    per = (len(z['src']) // (world * B)) * B
    lo = rank * per  # this rank's time chunk
    for s in range(n_steps):
        sl = slice(lo + s * B, lo + (s + 1) * B)
        tr.step(*(z[k][sl] for k in ('src', 'dst', 'neg', 'ts', 'eids')))
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu().numpy()
    np.savez(os.path.join(out_dir, f'fddp{rank}.npz'), params=flat, losses=tr.buf.losses.cpu().numpy())
    tdist.destroy_process_group()


@pytest.mark.gpu
def test_fused_trainer_data_parallel_two_ranks(tmp_path):
    world, n_steps = 2, 4
    mp.spawn(_fused_ddp_worker, args=(world, free_port(), n_steps, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(os.path.join(str(tmp_path), f'fddp{r}.npz')) for r in range(world))
    np.testing.assert_array_equal(r0['params'], r1['params'])  # same averaged gradients, same Adam state
    assert not np.allclose(r0['losses'], r1['losses'])           # different chunks
    # against a single process that averages the two chunks' gradients itself
    from test_hip_parity import build_hip_model
    from www2023tiger_amd.model.training import FusedTrainer, TrainBuffers
    z = load('train_seq_lr_d8')
    cfg = parse_cfg(z)
    B = cfg['B']
    per = (len(z['src']) // (world * B)) * B
    ma, _, _ = build_hip_model(z, cfg, dropout=0.0)
    mb, _, _ = build_hip_model(z, cfg, dropout=0.0)
    ma.train(); mb.train()
    ta = FusedTrainer(ma, B, lr=cfg['lr'], mutual=True)
    tb = TrainBuffers(mb, B, mutual=True)
    to = lambda x, dt: torch.as_tensor(x).to('cuda:0', dt)
    for s in range(n_steps):
        with torch.no_grad():  # model b follows model a's parameters; only its memories are its own
            for pa, pb in zip(ma.parameters(), mb.parameters()):
                pb.copy_(pa)
        a = [z[k][s * B:(s + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        b = [z[k][per + s * B:per + (s + 1) * B] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        tb.sb.load(to(b[0], torch.int64), to(b[1], torch.int64), to(b[2], torch.int64), to(b[3], torch.float64),
                   to(b[4], torch.int64))
        tb.launch()
        ta.load(*a)
        ta.buf.launch()
        ta.buf.gflat.add_(tb.gflat).mul_(0.5)
        ta.buf.flags.copy_(torch.maximum(ta.buf.flags, tb.flags))
        from www2023tiger_amd._lib import check, lib, ptr
        from www2023tiger_amd.hip_ops import stream_ptr
        check(lib.tg_adam_step(ptr(ta.segs), ta.n_segs, 4, ptr(ta.buf.flags), ptr(ta.steps), ta.lr, 0.9, 0.999, 1e-8,
                               1.0, stream_ptr(ma.device)), 'adam')
    ref = torch.cat([p.detach().flatten() for p in ma.parameters()]).cpu().numpy()
    assert rel_err(r0['params'], ref) < 1e-5
