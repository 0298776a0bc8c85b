"""CPU-only checks of the boundary: the library loads, exports every symbol the header
declares, and the host-side T-CSR build matches the oracle's graph."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'tiger_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tg_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from www2023tiger_amd import _lib
    names = header_functions()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f'{n} declared in tiger_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature'
    assert set(_lib.SIGNATURES) == set(names)
    assert _lib.lib.tg_abi_version() == 9


def test_struct_layouts_match_header():
    """ctypes mirrors must have the C struct sizes (all-8-byte fields after the int32 block)."""
    from www2023tiger_amd import _lib
    assert ctypes.sizeof(_lib.TgTcsr) == 6 * 8
    assert ctypes.sizeof(_lib.TgLinear) == 16
    assert ctypes.sizeof(_lib.TgModel) == 8 + 8 * 4 + 8 * 13 + 2 * 16 + 4 * 8 + 2 * 16 + 4 * 8 + 3 * 16 + 8 + 8 + 8 + 8 + 8
    assert ctypes.sizeof(_lib.TgStepIo) == 30 * 8


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from www2023tiger_amd import _lib
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.TigerHipError, match='no CPU fallback'):
        _lib._load()


def test_tcsr_host_build_matches_oracle():
    from oracle import tiger_oracle as O
    from www2023tiger_amd.data.graph import Graph
    rs = np.random.RandomState(0)
    E, n = 5000, 97
    src = rs.randint(1, 60, E)
    dst = rs.randint(60, n, E)
    ts = np.floor(rs.uniform(0, 300, E))  # NOT sorted: exercises the per-node stable sort
    eids = np.arange(1, E + 1)
    g = Graph.from_arrays(src, dst, ts, eids, strategy='recent_edges')
    o = O.OracleGraph(src, dst, ts, eids)
    assert g.num_node == o.num_node
    np.testing.assert_array_equal(g._h_indptr, o.indptr)
    np.testing.assert_array_equal(g._h_ts, o.ts)
    np.testing.assert_array_equal(g._h_nbr, o.nbr)
    np.testing.assert_array_equal(g._h_eid.view(np.uint32) & 0x7FFFFFFF, o.eid)
    np.testing.assert_array_equal(g._h_eid.view(np.uint32) >> 31, o.dir)
    # the adjacency-list constructor (reference signature) builds the same arrays
    adj = [[] for _ in range(g.num_node)]
    for s, d, t, e in zip(src, dst, ts, eids):
        adj[s].append((d, e, t, 0))
        adj[d].append((s, e, t, 1))
    g2 = Graph(adj, strategy='recent_edges')
    for a in ('_h_indptr', '_h_ts', '_h_nbr', '_h_eid'):
        np.testing.assert_array_equal(getattr(g, a), getattr(g2, a))


def test_tcsr_build_rejects_bad_ids():
    from www2023tiger_amd import _lib
    from www2023tiger_amd.data.graph import Graph
    with pytest.raises(ValueError):  # checked before either builder runs
        Graph.from_arrays(np.array([1, 2]), np.array([3, 4]), np.array([0.0, 1.0]), np.array([1, 2 ** 31]))
    with pytest.raises(ValueError):
        Graph.from_arrays(np.array([1, 9]), np.array([3, 4]), np.array([0.0, 1.0]), np.array([1, 2]), max_node_id=5)
    # the C entry point itself refuses them too
    bad = [np.array([1, 2]), np.array([3, 4]), np.array([0.0, 1.0]), np.array([1, 2 ** 31])]
    out = [np.empty(6, dtype=np.int64), np.empty(4), np.empty(4, dtype=np.int32), np.empty(4, dtype=np.int32)]
    rc = _lib.lib.tg_tcsr_build_host(2, *(_lib.ptr(a) for a in bad), 5, *(_lib.ptr(a) for a in out))
    assert rc == _lib.TG_EINVAL


def test_restart_run_entry_points_refuse_bad_arguments_before_any_device_call():
    """tg_eval_restart_run / tg_restart_seq_lists_fwd / tg_restart_static_lists_fwd (round 5): argument checks come first -
    callable here, without a GPU."""
    import ctypes as C
    from www2023tiger_amd import _lib
    lib = _lib.lib
    run = _lib.TgRestartRun()
    m, g = _lib.TgModel(), _lib.TgTcsr()
    assert lib.tg_eval_restart_run(None, None, None, None, None, 0, C.byref(run), 3, None) == _lib.TG_EINVAL
    io = _lib.TgTrainIo()
    assert lib.tg_eval_restart_run(C.byref(m), C.byref(g), None, C.addressof(io), None, 0, C.byref(run), 0, None) == _lib.TG_OK
    run.group = 0  # (no group size, no contexts, no row sets)
    assert lib.tg_eval_restart_run(C.byref(m), C.byref(g), None, C.addressof(io), None, 0, C.byref(run), 2, None) == _lib.TG_EINVAL
    run.group = 9  # beyond TG_RESTART_MAX_LISTS
    assert lib.tg_eval_restart_run(C.byref(m), C.byref(g), None, C.addressof(io), None, 0, C.byref(run), 2, None) == _lib.TG_EINVAL
    rs = _lib.TgSeqRestarter()
    for n_lists in (0, 9):
        assert lib.tg_restart_seq_lists_fwd(C.byref(m), C.byref(g), C.byref(rs), n_lists, None, None, None, None, None, None,
                                            None, None, 0, None) == _lib.TG_EINVAL
    assert lib.tg_restart_static_lists_fwd(C.byref(m), C.byref(g), None, None, 1, None, None, None, None, None, None, None,
                                           None) == _lib.TG_EINVAL


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every struct, as gcc lays out include/tiger_hip.h, against the ctypes mirrors."""
    import subprocess
    from www2023tiger_amd import _lib
    root = ROOT
    pairs = {'tg_tcsr': _lib.TgTcsr, 'tg_linear': _lib.TgLinear, 'tg_model': _lib.TgModel,
             'tg_seq_restarter': _lib.TgSeqRestarter, 'tg_step_io': _lib.TgStepIo,
             'tg_writeback_io': _lib.TgWritebackIo, 'tg_score_params': _lib.TgScoreParams,
             'tg_train_io': _lib.TgTrainIo, 'tg_adam_seg': _lib.TgAdamSeg, 'tg_lazy_restart': _lib.TgLazyRestart,
             'tg_part': _lib.TgPart, 'tg_restart_run': _lib.TgRestartRun}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "tiger_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(out[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(out[f'{cname}.{fname}']) == getattr(cls, fname).offset, f'{cname}.{fname}'


@pytest.mark.parametrize('name', ['ckpt_seq_lr_d8', 'ckpt_static_ll_d16'])
def test_state_dict_key_set_is_the_references(name):
    """The FULL key set (and order, shapes, dtypes) of a state_dict written by the reference - alias keys
    msg_memory.* / upd_memory.* and the duplicated time encoders included, mailbox and feature tables absent
    (non-persistent) - against this package's model on CPU tensors; strict loading succeeds."""
    import numpy as np
    import torch
    from _util import load, parse_cfg
    from test_hip_parity import _model_for_checkpoint
    z = load(name)
    cfg = parse_cfg(z)
    model, _, _ = _model_for_checkpoint(z, cfg, torch.device('cpu'))
    own = model.state_dict()
    keys = [str(k) for k in z['sd_keys']]
    assert list(own.keys()) == keys
    for k in keys:
        assert tuple(own[k].shape) == z['sd.' + k].shape, k
        assert own[k].numpy().dtype == z['sd.' + k].dtype, k
    res = model.load_state_dict({k: torch.from_numpy(z['sd.' + k]) for k in keys}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    np.testing.assert_array_equal(model.left_memory.vals.numpy(), z['sd.left_memory.vals'])
    assert model.msg_memory is (model.left_memory if cfg['msg_src'] == 'left' else model.right_memory)


def test_graft_entry_module_imports_and_builds():
    """the driver's entry points: the module must import (a syntax slip there would fail the round's build check) and
    build() must leave a library with the current ABI"""
    import importlib
    import __graft_entry__ as g
    importlib.reload(g)
    g.build()
    assert callable(g.smoke)
