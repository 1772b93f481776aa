"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/efe_engine.h declares; the Python mirror exposes the reference's method names."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'efe_engine.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(efe_[a-z_0-9]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from daimc_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} declared in efe_engine.h but not exported'
    assert sorted(_lib.EXPORTS) == names


def test_create_fails_cleanly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from daimc_amd import _lib
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    assert lib.efe_create(ctypes.byref(ctx), 0) != 0
    import daimc_amd
    with pytest.raises(RuntimeError):
        daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0)


def test_python_mirror_has_reference_api():
    import daimc_amd
    M = daimc_amd.ActiveInferenceModel
    for name in ('calculate_G', 'calculate_G_mean', 'calculate_G_repeated', 'calculate_G_4_repeated',
                 'calculate_G_given_trajectory', 'mcts_step_simulate', 'check_reward', 'habitual_net',
                 'imagine_future_from_o', 'save_weights', 'load_weights', 'save_all', 'load_all'):
        assert callable(getattr(M, name))
    for name in ('encoder', 'encoder_with_sample', 'decoder', 'reparameterize'):
        assert callable(getattr(daimc_amd.ModelDown, name))
    for name in ('transition', 'transition_with_sample', 'reparameterize'):
        assert callable(getattr(daimc_amd.ModelMid, name))
    assert callable(daimc_amd.ModelTop.encode_s)
    p = daimc_amd.MCTS_Params()
    assert (p.C, p.threshold, p.repeats, p.simulation_repeats, p.simulation_depth, p.use_habit, p.use_means) == \
        (1.0, 0.5, 300, 1, 3, False, True)


def test_host_softmax_matches_golden(golden):
    import numpy as np
    from daimc_amd import softmax_multi_with_log
    g = golden('rollout_m8d2s2')
    P, logP = softmax_multi_with_log(-g['sum_G'], 4)
    np.testing.assert_allclose(P, g['Ppi'], rtol=1e-6)
    np.testing.assert_allclose(logP, g['logPpi'], rtol=1e-6, atol=1e-6)


def test_torch_ops_library_registers_every_op():
    """torch.ops.efe.* (SURVEY 8b item 1): the registration library builds, loads without a GPU and carries the schemas;
    no CPU kernel exists, so a CPU call is a loud NotImplementedError (no fallback)."""
    import torch
    import __graft_entry__ as ge
    ge.build()
    from daimc_amd import _lib
    ops = _lib.load_ops()
    want = {'transition', 'decoder', 'encoder', 'habit', 'calculate_g', 'rollout', 'trajectory', 'simulate', 'action_posterior',
            'check_reward', 'reparameterize'}
    for n in want:
        schema = str(getattr(ops, n).default._schema)
        assert schema.startswith(f'efe::{n}(int ctx, Tensor'), schema
    assert 'Tensor? eps' in str(ops.calculate_g.default._schema) and 'int row_offset' in str(ops.rollout.default._schema)
    with pytest.raises(NotImplementedError):
        ops.habit(1, torch.zeros(2, 10))


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """a binary whose embedded source digest differs from the sources next to it must not load"""
    from daimc_amd import _lib, build
    assert build._stamp(_lib.LIB_PATH, 'EFE_BUILD_ID') == build.source_digest()
    monkeypatch.setattr(build, 'source_digest', lambda files=None: '0' * 16)
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(ImportError, match='stale'):
        _lib.load()


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` without a launcher environment starts its own N ranks -- one per GPU; with fewer visible HIP devices
    than ranks it stops with a message (exit code 2) instead of stacking ranks on one device (--share-device is the explicit
    launcher-test mode).  Runs on the CPU box: zero devices are visible here."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('two devices are visible: the launcher would start the ranks')
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert 'one rank per GPU' in r.stderr and r.stdout.strip() == ''
