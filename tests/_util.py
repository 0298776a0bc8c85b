"""Helpers shared by the CPU (oracle) and GPU (parity) tests."""
import ast
import os

import numpy as np

from golden._weights import golden_param

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MODEL_FIXTURES = ['seq_lr_d8', 'static_ll_d16', 'seq_rr_d8_nofeat', 'static_ll_d172', 'seq_lr_d172',
                  'mlp_merge_d8', 'linear_gru_d8']
TWO_LAYER_FIXTURES = ['static_lr_d8_L2', 'seq_ll_d16_L2']  # --n_layers 2 (hop-2 sampled at the neighbours' timestamps)


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def parse_cfg(z):
    cfg = {}
    for kv in z['cfg']:
        k, v = str(kv).split('=', 1)
        try:
            cfg[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            cfg[k] = v
    return cfg


def fixture_params(z, cfg):
    names = [str(x) for x in z['param_names']]
    shapes = [tuple(int(t) for t in str(s).split(',')) for s in z['param_shapes']]
    return {n: golden_param(n, s, cfg['wseed']) for n, s in zip(names, shapes)}


def fixture_tables(z, cfg):
    n_nodes = int(z['n_nodes'])
    d = cfg['d']
    if 'nfeats' in z.files:
        nfeats = z['nfeats']
    elif cfg.get('nfeat') == 'zero':
        nfeats = np.zeros((n_nodes, d), dtype=np.float32)
    else:
        nfeats = None
    efeats = z['efeats'] if 'efeats' in z.files else None
    return n_nodes, nfeats, efeats


def n_batches(z):
    return max(int(k[1:].split('_')[0]) for k in z.files if k[0] == 'b' and k[1].isdigit()) + 1


def rel_err(a, b):
    """max |a-b| / max(1, max|b|): the 'relative fp32' measure used for embeddings."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0 and b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def row_rel_err(a, b, floor=1e-3):
    """Truly relative measure: max over rows of ||a_i - b_i||_2 / ||b_i||_2, taken over the rows whose
    reference norm exceeds `floor`; rows below the floor (padding / never-written rows) must agree to
    `floor` * 1e-4 absolutely, reported on the same scale."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0 and b.size == 0:
        return 0.0
    a = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a.reshape(1, -1)
    b = b.reshape(a.shape)
    nb = np.sqrt((b * b).sum(1))
    nd = np.sqrt(((a - b) ** 2).sum(1))
    big = nb > floor
    worst = float((nd[big] / nb[big]).max()) if big.any() else 0.0
    if (~big).any():
        worst = max(worst, float(nd[~big].max() / floor))
    return worst


WORST = {}  # test id -> {what: (max-abs-relative, row-relative)}; printed by conftest.pytest_terminal_summary


def assert_close(a, b, what, tol=1e-4, row_tol=None):
    """Both parity measures for float32 results: max|a-b| / max(1, max|b|) < tol AND the per-row relative
    L2 error < row_tol (default tol).  The worst values of every test are collected for the session report."""
    row_tol = tol if row_tol is None else row_tol
    e1, e2 = rel_err(a, b), row_rel_err(a, b)
    test = os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0].split('::', 1)[-1]
    cur = WORST.setdefault(test, {}).get(what, (0.0, 0.0))
    WORST[test][what] = (max(cur[0], e1), max(cur[1], e2))
    assert e1 < tol, (what, 'max-abs-relative', e1)
    assert e2 < row_tol, (what, 'row-relative', e2)
    return e1, e2
