"""Input side: load_jodie_data splits, negative-sample streams and the
DDP ChunkSampler against what the reference produced on the same toy files
(tests/golden/input_side.npz).  CPU only."""
import os
import sys

import numpy as np
import pytest

from _util import load

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))


def write_files(root, name, z, with_feats=True):
    import pandas as pd
    os.makedirs(os.path.join(root, 'data'), exist_ok=True)
    n = len(z['src'])
    df = pd.DataFrame({'u': z['src'], 'i': z['dst'], 'ts': z['ts'], 'label': z['labels'], 'idx': np.arange(1, n + 1)})
    df.to_csv(os.path.join(root, 'data', f'ml_{name}.csv'))
    if with_feats:
        np.save(os.path.join(root, 'data', f'ml_{name}.npy'), z['efeats'])
        np.save(os.path.join(root, 'data', f'ml_{name}_node.npy'), z['nfeats'])


def test_load_jodie_data_matches_reference(tmp_path):
    from www2023tiger_amd.data.data_loader import load_jodie_data
    z = load('input_side')
    write_files(str(tmp_path), 'toy', z)
    names = ('full', 'train', 'val', 'test', 'ind_val', 'ind_test')
    for seed in (0, 7):
        res = load_jodie_data('toy', train_seed=seed, root=str(tmp_path))
        np.testing.assert_array_equal(res[0], z['nfeats'])
        np.testing.assert_array_equal(res[1], z['efeats'])
        for nm, dset in zip(names, res[2:]):
            np.testing.assert_array_equal(np.asarray(dset.eids), z[f's{seed}_{nm}_eids'], err_msg=nm)
            if f's{seed}_{nm}_neg' in z.files:
                np.testing.assert_array_equal(np.asarray(dset.neg_dst), z[f's{seed}_{nm}_neg'], err_msg=nm)
        tr = res[3]
        np.testing.assert_array_equal(np.array([tr[i][2] for i in range(50)]), z[f's{seed}_train_draws'])
    write_files(str(tmp_path), 'bare', z, with_feats=False)
    res = load_jodie_data('bare', train_seed=1, root=str(tmp_path), val_p=0.6, test_p=0.8)
    assert res[0] is None and res[1] is None
    np.testing.assert_array_equal(np.asarray(res[3].eids), z['bare_train_eids'])
    np.testing.assert_array_equal(np.asarray(res[7].eids), z['bare_ind_test_eids'])


def test_chunk_sampler_matches_reference():
    from www2023tiger_amd.data.data_loader import ChunkSampler
    z = load('input_side')
    for n, rank, ws, bs, seed, epoch, length, first, last in z['chunk_sampler'].tolist():
        cs = ChunkSampler(n, rank, ws, bs, seed)
        cs.set_epoch(epoch)
        idx = list(iter(cs))
        assert len(cs) == length == len(idx)
        assert (idx[0], idx[-1]) == (first, last)
        assert idx == list(range(first, last + 1))


def test_small_helpers():
    from www2023tiger_amd.data.data_loader import compute_delta_std, is_sorted
    assert is_sorted([1, 1, 2, 5]) and not is_sorted([1, 3, 2])
    src, dst, ts = np.array([1, 2, 1]), np.array([3, 3, 2]), np.array([1.0, 4.0, 6.0])
    # deltas: (1-0, 1-0), (4-0, 4-1), (6-1, 6-4)
    assert abs(compute_delta_std(src, dst, ts) - np.std([1, 1, 4, 3, 5, 2])) < 1e-12


def test_native_negative_stream_equals_per_event_draws():
    """RandEdgeSampler.sample_pairs(n) == n calls of sample(1): same values, same final RandomState."""
    from www2023tiger_amd.data.data_loader import RandEdgeSampler
    rs = np.random.RandomState(3)
    for n_src, n_dst, n in ((7, 1, 50), (1000, 33, 700), (2 ** 20 + 3, 5, 200), (1, 1, 10)):
        src_list = np.sort(rs.choice(10 ** 7, n_src, replace=False))
        dst_list = np.sort(rs.choice(10 ** 7, n_dst, replace=False)) + 10 ** 7
        a = RandEdgeSampler(src_list, dst_list, seed=11)
        b = RandEdgeSampler(src_list, dst_list, seed=11)
        a.sample(3), b.sample(3)  # both streams already advanced (position inside the key block)
        s_ref = np.array([a.sample(1) for _ in range(n)]).reshape(n, 2)
        s_got, d_got = b.sample_pairs(n)
        np.testing.assert_array_equal(s_got, s_ref[:, 0])
        np.testing.assert_array_equal(d_got, s_ref[:, 1])
        np.testing.assert_array_equal(a.sample(5)[1], b.sample(5)[1])  # states agree afterwards


def test_interaction_data_get_batch_equals_getitem():
    from www2023tiger_amd.data.data_loader import InteractionData
    z = load('input_side')
    lab = z['labels']
    eids = np.arange(1, len(lab) + 1)
    for ev in (False, True):
        a = InteractionData(z['src'], z['dst'], z['ts'], eids, lab, seed=5, eval=ev)
        b = InteractionData(z['src'], z['dst'], z['ts'], eids, lab, seed=5, eval=ev)
        items = [a[i] for i in range(40, 140)]
        got = b.get_batch(40, 140)
        for col in range(6):
            np.testing.assert_array_equal(np.array([it[col] for it in items]), got[col])


def test_optim_adam_plain_path_equals_torch_adam():
    """www2023tiger_amd.optim.Adam on gradients from ordinary autograd (no fused train step): the
    torch-op fallback follows torch.optim.Adam step for step, and parameters without a gradient are
    skipped (no state, no step count), as torch does."""
    import torch
    from www2023tiger_amd.optim import Adam
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa, ob = Adam(a, lr=3e-3, betas=(0.8, 0.95), eps=1e-6), torch.optim.Adam(b, lr=3e-3, betas=(0.8, 0.95), eps=1e-6)
    assert all(getattr(p, '_tg_deferred', False) for p in a)
    for it in range(6):
        for ps, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            x = torch.arange(15, dtype=torch.float32).reshape(5, 3) * 0.1 + it
            loss = ((ps[0] * x).sum() - 1.0) ** 2 + (ps[1] ** 3).sum() * (it % 2)  # ps[1] idle on even iterations
            loss.backward()
            if it % 2 == 0:
                ps[1].grad = None
            opt.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-7)
    assert a[2].grad is None and 'exp_avg' not in oa.state[a[2]]
    assert oa.state[a[1]]['step'] == 3 and oa.state[a[0]]['step'] == 6


@pytest.mark.gpu
def test_device_negative_stream_continues_the_datasets_random_state():
    """tg_rand_edge_pairs: the per-event two-draw RandomState stream of RandEdgeSampler.sample(1) on the GPU
    (data_loader.py:246-251,291-294), bit for bit - against the reference's own draws (input_side.npz), against
    the host routine on awkward table sizes (powers of two, size 1, sizes whose mask rejects half the words),
    across state-block boundaries, and interleaved with host draws of the same sampler."""
    import torch
    from www2023tiger_amd.data.data_loader import InteractionData, RandEdgeSampler
    dev = torch.device('cuda', 0)
    z = load('input_side')
    for seed in (0, 7):   # the reference's training split draws negatives from ITS OWN node lists, seed = train_seed
        eids = z[f's{seed}_train_eids'] - 1
        tr = InteractionData(z['src'][eids], z['dst'][eids], z['ts'][eids], z[f's{seed}_train_eids'], z['labels'][eids],
                             seed=seed, eval=False)
        got = tr.neg_dst_sampler.sample_pairs_device(50, dev)[1].cpu().numpy()
        np.testing.assert_array_equal(got, z[f's{seed}_train_draws'])
    rs = np.random.RandomState(5)
    for n_src, n_dst in ((1, 1), (1, 37), (64, 1), (64, 64), (33, 65), (1000, 3), (5, 100000), (129, 2 ** 20 + 1)):
        a = RandEdgeSampler(np.arange(n_src) * 3 + 1, np.arange(n_dst) * 2 + 5, seed=11)
        b = RandEdgeSampler(np.arange(n_src) * 3 + 1, np.arange(n_dst) * 2 + 5, seed=11)
        for count in (1, 7, 300, 2048, 1):                        # 2048 pairs cross several 624-word state blocks
            hs, hd = a.sample_pairs(count)
            ds, dd = b.sample_pairs_device(count, dev)
            np.testing.assert_array_equal(ds.cpu().numpy(), hs, err_msg=f'{n_src}x{n_dst} src')
            np.testing.assert_array_equal(dd.cpu().numpy(), hd, err_msg=f'{n_src}x{n_dst} dst')
        # host draws continue the device stream and the other way round
        np.testing.assert_array_equal(b.sample(5)[1], a.sample(5)[1])
        np.testing.assert_array_equal(b.sample_pairs_device(9, dev)[1].cpu().numpy(), a.sample_pairs(9)[1])
        assert a.rng.get_state()[2] == b.rng.get_state()[2] and (a.rng.get_state()[1] == b.rng.get_state()[1]).all()


@pytest.mark.gpu
def test_device_resident_batches_equal_the_host_loader(tmp_path):
    """BatchLoader over a dataset kept on the GPU (InteractionData.to_device): ids, times, labels and the
    on-the-fly negatives of every batch equal those of the host loader; nothing but launches happens per batch."""
    import torch
    from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, load_jodie_data
    from www2023tiger_amd.data.graph import Graph
    dev = torch.device('cuda', 0)
    z = load('input_side')
    write_files(str(tmp_path), 'toy', z)
    outs = []
    for on_device in (False, True):
        res = load_jodie_data('toy', train_seed=3, root=str(tmp_path))
        full, train = res[2], res[3]
        g = Graph.from_data(train, strategy='recent_edges', seed=0, max_node_id=int(max(full.src.max(), full.dst.max())),
                            device=dev)
        coll = GraphCollator(g, 4, 1, restarter='static')
        if on_device:
            train.to_device(dev)
        rows = []
        for s, d, n, t, e, lab, cg in BatchLoader(train, 64, coll):
            rows.append([x.cpu().numpy() for x in (s, d, n, t, e, lab, cg.ts64, cg.layers[1][0])])
        outs.append(rows)
    assert len(outs[0]) == len(outs[1]) > 3
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
