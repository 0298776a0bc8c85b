"""Every kernel-selection knob of the library (the knob table) under its non-default values: the C2 timed form -
resident stream, lean eager step, collate prefetch, graph of several steps - still matches the oracle (VERDICT r03 task 8).
The library reads a knob once per process, so each setting runs tests/_knob_case.py in a child process (one at a time)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

SETTINGS = [
    # the merged fc1 product: K-split blocks (default) -> stream-K pieces -> plain blocks; eight wavefronts per block
    'TG_GEMM_KS16=0', 'TG_GEMM_KS16=8', 'TG_GEMM_KS16=0 TG_GEMM_SK=0', 'TG_GEMM_KS16=0 TG_SK_WORKERS=384',
    'TG_GEMM_KS16=0 TG_GEMM_ASK_KS=1', 'TG_GEMM_KS16=0 TG_GEMM_ASK_KS=1 TG_GEMM_ASK_DEPTH=4', 'TG_GEMM_KS16_TILES=100',
    # short-K products (fc2, query rows): LDS-free -> activation-stationary -> 64 x 64 blocks (one / two k-groups, depth 4)
    'TG_GEMM_DIRECT=0', 'TG_GEMM_DIRECT=0 TG_GEMM_ASTAT=0', 'TG_GEMM_DIRECT=0 TG_GEMM_ASTAT=0 TG_GEMM_DEPTH=4',
    'TG_GEMM_DIRECT=0 TG_GEMM_ASTAT=0 TG_GEMM_KS=2', 'TG_GEMM_DIRECT=0 TG_GEMM_ASTAT=4', 'TG_GEMM_DIRECT=0 TG_GEMM_ASTAT_CPB=1',
    # riders and the forms they need
    'TG_WB_RIDER=0', 'TG_WB_RIDER_FC1=0', 'TG_PREFETCH=0', 'TG_PREFETCH_SPLIT=1', 'TG_CTAB=0', 'TG_GTAB=0', 'TG_GTAB=0 TG_ATTN_TILE=1',
    'TG_EAGER_DIRECT=0',
    # the updater: 16-column LDS-free blocks (default) -> 32-row LDS-free -> LDS-staged 32 / 64 / 96 / 128-row blocks
    'TG_GRU_D16=0', 'TG_GRU_D16=0 TG_GRU_DIRECT=0', 'TG_GRU_D16=0 TG_GRU_MICRO=0', 'TG_GRU_D16=0 TG_GRU_NW=3',
    'TG_GRU_D16=0 TG_GRU_NW=4', 'TG_GRU_D16=0 TG_GRU_NW=4 TG_GRU_KS=1', 'TG_GRU_D16=0 TG_GRU_NW=4 TG_GRU_KS=2', 'TG_GRU_D16=3',
    # the split updater (off by default)
    'TG_GRU_SPLIT=1', 'TG_GRU_SPLIT=1 TG_KS16_SECOND=0', 'TG_GRU_SPLIT=2', 'TG_GRU_SPLIT=2 TG_GRU_SPLIT_BOX=0',
    # fc1 + fc2 as one launch of 16-row blocks (off by default)
    'TG_FC12=1', 'TG_FC12=1 TG_WB_RIDER=0',
]


GROUP = 4  # child processes at a time


def _run_case(setting):
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split('=')
        env[k] = v
    env.setdefault('OMP_NUM_THREADS', '4')  # (the children's oracle steps would otherwise each take every core)
    env.setdefault('MKL_NUM_THREADS', '4')
    r = subprocess.run([sys.executable, os.path.join(HERE, '_knob_case.py')], env=env, capture_output=True, text=True,
                       timeout=600)
    return r.returncode, r.stdout, r.stderr


_RESULTS = {}


def _result(setting):
    """The library reads a knob once per process, so every setting needs a process of its own; run one after the other
    they were half of the suite's wall time.  The first test of every group of GROUP settings runs the whole group, GROUP
    child processes at a time - they share the test box's one GPU, within its process guard, and each is a short chain
    of small launches."""
    if setting not in _RESULTS:
        from concurrent.futures import ThreadPoolExecutor
        i = SETTINGS.index(setting)
        group = SETTINGS[i - i % GROUP:i - i % GROUP + GROUP]
        with ThreadPoolExecutor(max_workers=GROUP) as ex:
            _RESULTS.update(zip(group, ex.map(_run_case, group)))
    return _RESULTS[setting]


@pytest.mark.gpu
@pytest.mark.parametrize('setting', SETTINGS, ids=[s.replace(' ', ',') for s in SETTINGS])
def test_c2_timed_form_under_knob(setting):
    rc, out, err = _result(setting)
    tail = (out + err)[-3000:]
    assert rc == 0 and 'KNOB-CASE OK' in out, f'{setting}:\n{tail}'


@pytest.mark.gpu
@pytest.mark.parametrize('setting', ['TG_TRAIN_SIDE=0', 'TG_TRAIN_FORK=0', 'TG_SEQ_BWD_SPLIT=0'])
def test_training_step_under_lane_knobs(setting):
    """The training step's second stream (knob table): without the lane, and with the lane forked behind the whole
    forward pass instead of right behind the sampler - the mutual-loss gradient tests (fixtures, C2 widths, the reference's
    trajectory) in a child process that reads the knob."""
    env = dict(os.environ)
    k, v = setting.split('=')
    env[k] = v
    env.setdefault('OMP_NUM_THREADS', '4')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(HERE, 'test_hip_train.py'), '-q', '-x', '-m', 'gpu', '-k',
                        'mutual_gradients_match_oracle or c2_width or mutual_trajectory', '-p', 'no:cacheprovider'],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and ' passed' in r.stdout, (r.stdout + r.stderr)[-3000:]
