import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The shared library is a build product (git-ignored): make sure it exists and is not older than its
    sources before anything imports the package (hipcc cross-compiles gfx950 without a GPU)."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, 'www2023tiger_amd', 'csrc')
    lib = os.path.join(csrc, 'libtiger_hip.so')
    srcs = glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')) + \
        [os.path.join(ROOT, 'include', 'tiger_hip.h')]
    if os.path.exists(lib) and os.path.getmtime(lib) >= max(os.path.getmtime(f) for f in srcs):
        return
    if os.path.exists('/opt/rocm/bin/hipcc'):
        subprocess.run(['make', '-j4', '-C', csrc], check=True, stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so that
    `pytest tests/` without a marker expression works in the CPU container."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    """Worst parity errors per test (max|a-b|/max(1,max|ref|) and per-row ||a-b||/||ref||)."""
    try:
        from _util import WORST
    except Exception:
        return
    if not WORST:
        return
    terminalreporter.write_line('parity, worst error per test  [max-abs-relative | row-relative]')
    for test, d in WORST.items():
        e1 = max(v[0] for v in d.values())
        e2 = max(v[1] for v in d.values())
        which = max(d, key=lambda k: d[k][1])
        terminalreporter.write_line(f'  {test}: {e1:.2e} | {e2:.2e} ({which})')
