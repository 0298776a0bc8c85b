"""Randomised shapes through the fused step (C ABI) against the CPU oracle: widths that are not multiples of 32,
neighbour counts from 1 to 16, every msg_src / upd_src pairing, with and without edge / node feature tables, batch sizes
from a handful to a few hundred, in the three forms of the step (lazy updater, eager, eager + lean) and with the
attention weights as stored or pre-multiplied.  Small graphs, so the oracle stays fast; the point is the variety of
tile / tail / block-selection edges the kernels see (k_gru_direct, k_gemm_astat, the two-/four-column core, lean
dedup slots), not size.  Indices bit-exact, float32 within 1e-4 under both measures of _util.assert_close."""
import numpy as np
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _cases():
    rs = np.random.RandomState(2024)
    out = []
    dims = [8, 12, 20, 44, 64, 100, 128, 172, 200, 256]
    for i in range(28):
        d = int(dims[rs.randint(len(dims))])
        out.append(dict(seed=i, d=d, K=int(rs.choice([1, 2, 5, 10, 16])), B=int(rs.choice([3, 17, 64, 200, 333])),
                        n_u=int(rs.choice([30, 150, 700])), n_i=int(rs.choice([10, 40, 200])),
                        msg_src=str(rs.choice(['left', 'right'])), upd_src=str(rs.choice(['left', 'right'])),
                        efeats=bool(rs.rand() < 0.6), nfeats=bool(rs.rand() < 0.5), fuse=bool(rs.rand() < 0.6),
                        form=str(rs.choice(['lazy', 'eager', 'lean', 'lean'])), integer_ts=bool(rs.rand() < 0.7)))
    return out


CASES = _cases()


@pytest.mark.parametrize('c', CASES, ids=[f"{c['form']}-d{c['d']}-K{c['K']}-B{c['B']}-{c['msg_src'][0]}{c['upd_src'][0]}"
                                          f"{'-e' if c['efeats'] else ''}{'-n' if c['nfeats'] else ''}{'-f' if c['fuse'] else ''}"
                                          for c in CASES])
def test_random_configuration_matches_oracle(c):
    import bench
    from oracle import tiger_oracle as O
    from test_hip_parity import compare_state_with_oracle
    nb = 7
    B, K, d = c['B'], c['K'], c['d']
    E = nb * B + 5
    stream = bench.make_stream(c['n_u'], c['n_i'], E, 3.0e4 * E / 1000.0, seed=c['seed'], d_e=d,
                               integer_ts=c['integer_ts'], with_efeats=c['efeats'])
    model, orc = bench.build_models(stream, d, K, c['msg_src'], c['upd_src'], with_oracle=True,
                                    zero_nfeats=c['nfeats'], seed=c['seed'])
    if c['nfeats']:  # non-zero node features on both sides (the JODIE tables are zeros; the code path is not)
        nf = np.random.RandomState(c['seed']).randn(stream['n_nodes'], d).astype(np.float32) * 0.3
        nf[0] = 0
        model.raw_feat_getter.nfeats.copy_(torch.from_numpy(nf))
        orc.nfeats = torch.from_numpy(nf)
        model._struct_cache = None
    if c['fuse']:
        model.fuse_attention()
    if c['form'] != 'lazy':
        model.eager_updates()
    for b in range(nb):
        lo, hi = b * B, (b + 1) * B + (5 if b == nb - 1 else 0)   # ragged last batch
        a = [stream[k][lo:hi] for k in ('src', 'dst', 'neg', 'ts', 'eids')]
        n = hi - lo
        buf = model.stream_step(*a, lean=(c['form'] == 'lean'))
        cg = O.collate(orc.graph, a[0], a[1], a[2], a[3], K, 'static')
        ref = orc.stream_step(*a, cg).numpy()
        np.testing.assert_array_equal(buf.l1_nids.cpu().numpy()[:3 * n], cg['l1_nids'])
        np.testing.assert_array_equal(buf.l1_eids.cpu().numpy()[:3 * n], cg['l1_eids'])
        assert_close(buf.h[:2 * n].cpu().numpy(), ref, 'h_left', TOL)
        if b == 3:  # a flush in the middle (tiger.py: flush_msg): pending messages consumed on both sides
            model.flush_msg()
            orc.flush_msg()
    compare_state_with_oracle(model, orc)
