"""Helpers shared by the CPU (oracle) and GPU (parity) tests."""
import ast
import os

import numpy as np

from golden._weights import golden_param

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MODEL_FIXTURES = ['seq_lr_d8', 'static_ll_d16', 'seq_rr_d8_nofeat', 'static_ll_d172', 'seq_lr_d172',
                  'mlp_merge_d8', 'linear_gru_d8']


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
