"""Seeded parameter sweep over the other kernels of the path: the shape-parametrised checks of tests/test_gn_step.py,
test_lm_solver.py, test_sweep_kernels.py, test_rslm.py and test_preprocess.py re-run at shapes no fixed parametrisation
lists (ragged N, one point per lane up to the streaming mode, few / many proposals, every bounds mode).  A few on the
emulator, the full list on the GPU."""
import os
import random

import pytest
import torch

import test_gn_step
import test_lm_solver
import test_preprocess
import test_rslm
import test_sweep_kernels


def _cases(n, seed, n_max):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        kind = ('gn_step', 'lm', 'normal_eq', 'rslm_oracle', 'rslm_composite', 'prepare', 'prepare_dense')[i % 7]
        dof = rng.choice((6, 4))
        bounds = rng.choice((None, 'tensor', 'tight'))
        N = min(rng.choice((9, 31, 64, 65, 127, 200, 333, 512, 513, 800, 1100, 2048, 3000, 8300)), n_max)
        if kind == 'gn_step':
            args = (dof, N, bounds)
        elif kind == 'lm':
            args = (dof, max(N, 33), rng.choice((2, 3, 5, 8)), rng.random() < 0.3)
        elif kind == 'normal_eq':
            args = (dof, bounds, rng.choice((1, 3, 10)), max(N, 16))
        elif kind == 'rslm_oracle':
            n = rng.choice((4, 8, 12, 16)) if dof == 4 else rng.choice((8, 12, 16))
            args = (dof, min(max(N, 2 * n), 512), rng.choice((4, 16, 40, 64)), n, bounds)
        elif kind == 'rslm_composite':
            n = rng.choice((4, 8, 16)) if dof == 4 else rng.choice((8, 12, 16))     # 4 points barely determine 6 DoF: chaotic
            args = (dof, min(max(N, 2 * n), 512), rng.choice((4, 20, 48)), n, bounds, rng.random() < 0.3)
        elif kind == 'prepare':
            args = (rng.choice(('softmax', 'mean_exp')), rng.random() < 0.6, rng.random() < 0.6, min(max(N, 16), 2048))
        else:
            H, W = rng.choice(((8, 8), (16, 12), (32, 32), (64, 64)))
            args = (rng.choice(('softmax', 'mean_exp')), rng.random() < 0.6, rng.random() < 0.6, H, W,
                    min(rng.choice((16, 100, 512)), H * W))      # pixels are sampled without replacement
        out.append(pytest.param(kind, args, id=f'{kind}-' + '-'.join(str(a) for a in args)))
    return out


def _run(backend, monkeypatch, kind, args):
    if kind == 'gn_step':
        test_gn_step.test_gn_step_forward_backward(backend, *args)
    elif kind == 'lm':
        test_lm_solver.test_lm_vs_oracle_seeded(backend, *args)
    elif kind == 'normal_eq':
        test_sweep_kernels.test_normal_equations_launch_shapes_agree(backend, monkeypatch, *args)
    elif kind == 'rslm_oracle':
        test_rslm.test_fused_rslm_matches_oracle_on_injected_draws(backend, *args)
    elif kind == 'rslm_composite':
        test_rslm.test_fused_matches_composite_on_same_draws(backend, monkeypatch, *args)
    elif kind == 'prepare':
        test_preprocess.test_prepare_matches_torch(backend, *args)
    else:
        test_preprocess.test_prepare_dense_matches_reference_composite(backend, *args)


@pytest.mark.parametrize('kind,args', _cases(14, 5, 600))
def test_kernel_sweep_small(backend, monkeypatch, poisoned_empty, kind, args):
    _run(backend, monkeypatch, kind, args)


@pytest.mark.gpu
@pytest.mark.parametrize('kind,args', _cases(int(os.environ.get('EPROPNP_FUZZ_CASES', '84')), int(os.environ.get('EPROPNP_FUZZ_SEED', '6')), 9000))
def test_kernel_sweep_gpu(monkeypatch, poisoned_empty, kind, args):
    import install as emu
    emu.uninstall()
    _run(torch.device('cuda:0'), monkeypatch, kind, args)
