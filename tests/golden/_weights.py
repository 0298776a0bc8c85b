"""Deterministic model weights shared by the golden-vector generator and the tests.

Fixtures store only inputs and expected outputs; the parameters are regenerated
from (name, shape, seed) with numpy's legacy MT19937 stream, which is stable
across numpy versions.  Both the reference model (in make_golden.py) and the
build's model (in tests/) load the same values by state_dict key.
"""
import zlib

import numpy as np


def golden_param(name: str, shape, seed: int) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'basis_freq':  # keep the TGAT initialiser (time_encoding.py:13)
        return (1.0 / 10 ** np.linspace(0, 9, shape[0])).astype(np.float32)
    if leaf == 'phase':
        return rs.uniform(-0.5, 0.5, shape).astype(np.float32)
    if len(shape) >= 2:
        parent = name.split('.')[-2] if '.' in name else ''
        if parent.endswith('_emb') or parent == 'hit_embedding':  # embedding tables
            return rs.uniform(-0.5, 0.5, shape).astype(np.float32)
        a = 1.0 / np.sqrt(shape[-1])
        return rs.uniform(-a, a, shape).astype(np.float32)
    return rs.uniform(-0.1, 0.1, shape).astype(np.float32)


def golden_state_dict(named_shapes, seed: int):
    """named_shapes: iterable of (name, shape) for *parameters* only."""
    return {n: golden_param(n, s, seed) for n, s in named_shapes}
