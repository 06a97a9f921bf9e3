"""The reference's own callers against the HIP path, where the reference checkout is absent (missing #1).

tests/golden/callers_{notebook,linemod_train,det_head}.npz hold what the UNMODIFIED reference computes when the literal source
of its callers is executed (oracle/make_golden.py -> oracle/run_callers.py, build container): demo/fit_identity.ipynb cells 5-10
+ forward_test, EPro-PnP-6DoF/lib/train.py:47-57,141-193, EPro-PnP-Det/.../deform_pnp_head.py:870-893,514-527 -- losses,
pose_opt_plus, the gradients that reach the network outputs, on seeded inputs and injected random draws.  Here the same slices,
restated (oracle/callers_restated.py: pinned bit for bit to the literal source on the reference by make_golden.py, and to the
literal source on the package by tests/test_reference_callers.py), run on the package -- on cuda:0 through libepropnp_hip.so
(`-m gpu`) and on the CPU emulation of the kernels -- with the same draws, and are compared with those fixtures at the bars of
tests/test_reference_callers.py."""
import os

import numpy as np
import pytest
import torch

import callers_restated
import run_callers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _fixture(scenario):
    d = dict(np.load(os.path.join(GOLDEN, run_callers.FIXTURES[scenario])))
    return d, int(d['meta.objects']), int(d['meta.steps'])


def _run(scenario, dev):
    ref, objects, steps = _fixture(scenario)
    threads = torch.get_num_threads()
    names, restore = run_callers.patch_package(run_callers.Draws(seed=run_callers.SEEDS[scenario]))
    try:
        torch.set_num_threads(4)          # (the CPU side of the slices -- data generation, the MLP on the emulation backend)
        out = callers_restated.SCENARIOS[scenario](names, dev, objects, steps)
    finally:
        restore()
        torch.set_num_threads(threads)
    pkg = {k: v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v) for k, v in out.items()}
    assert set(pkg) == {k for k in ref if not k.startswith(('spread.', 'meta.'))}
    return ref, pkg


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _lse(logw):
    m = logw.max(0)
    return m + np.log(np.exp(logw - m).sum(0))


def test_linemod_training_slice_against_the_reference(backend):
    """lib/train.py:141-193 on a dense-correspondence scene: 64 x 64 maps, 512 sampled pixels, tensor bounds, z_min 0.01,
    relative_delta 0.1, RSLM(16,4,3) + LM 5, S = 512, with_pose_opt_plus + derivative regularisation."""
    ref, pkg = _run('train6dof', backend)
    for it in range(2):
        for k in ('loss_mc', 'loss_t', 'loss_r', 'cost_tgt'):
            assert np.abs(pkg[f'step{it}.{k}'] - ref[f'step{it}.{k}']).max() <= 1e-6 + 2e-4 * np.abs(ref[f'step{it}.{k}']).max(), (it, k)
        assert np.abs(pkg[f'step{it}.pose_opt_plus'] - ref[f'step{it}.pose_opt_plus']).max() <= 2e-4
        assert np.abs(_lse(pkg[f'step{it}.pose_sample_logweights']) - _lse(ref[f'step{it}.pose_sample_logweights'].astype(np.float64))).max() <= 1e-3
        for k in ('g_noc', 'g_logit', 'g_scale'):
            assert _rel(pkg[f'step{it}.{k}'], ref[f'step{it}.{k}']) <= 2e-3, (it, k)
    assert abs(float(pkg['norm_factor_buffer']) - float(ref['norm_factor_buffer'])) <= 1e-6


def test_detection_head_slices_against_the_reference(backend):
    """deform_pnp_head.py:870-893 (two decoder stages through the Det loss module, then forward(with_pose_opt_plus=True)) with the
    layer built from the config dict, and :514-527 (test_post, both branches): EProPnP4DoF, normalize=True, img_shape bounds,
    RSLM(16,64,3) + LM 10."""
    ref, pkg = _run('det', backend)
    for it in range(2):
        for k in ('loss_pose_0', 'loss_pose_1', 'norm_factor'):
            assert _rel(pkg[f'step{it}.{k}'], ref[f'step{it}.{k}']) <= 2e-4, (it, k)
        for k in ('pose_opt', 'pose_opt_plus'):
            assert np.abs(pkg[f'step{it}.{k}'] - ref[f'step{it}.{k}']).max() <= 5e-4, (it, k)
        for k in ('g_noc0', 'g_noc1', 'g_w2d0', 'g_w2d1', 'g_scale'):
            assert _rel(pkg[f'step{it}.{k}'], ref[f'step{it}.{k}']) <= 2e-3, (it, k)
    assert np.abs(pkg['norm_factor_buffers'] - ref['norm_factor_buffers']).max() <= 1e-6
    assert np.abs(pkg['test_plain.pose_opt'] - ref['test_plain.pose_opt']).max() <= 1e-4
    assert np.abs(pkg['test_mc.pose_opt'] - ref['test_mc.pose_opt']).max() <= 1e-4
    assert np.abs(pkg['test_mc.pose_sample_weights'] - ref['test_mc.pose_sample_weights']).max() <= 1e-4
    assert np.abs(pkg['test_mc.pose_samples'] - ref['test_mc.pose_samples']).max() <= 1e-3


def test_notebook_cells_against_the_reference(backend):
    """demo/fit_identity.ipynb cells 5-10 (3 Adam steps of the untrained MLP) and forward_test in both solver modes.  An untrained
    network emits ill-conditioned correspondences -- the reference's own outputs move by ~1e-3 when its inputs move by 1 ulp --
    so the yardstick is `spread.*` of the fixture: the reference against itself under a 2e-7 relative jitter of the inputs."""
    ref, pkg = _run('notebook', backend)
    assert pkg['printed'].shape == (3, 6) and np.isfinite(pkg['printed']).all()
    for k in pkg:
        err = np.abs(pkg[k] - ref[k]).max()
        assert err <= 10 * float(ref['spread.' + k]) + 1e-3 * max(np.abs(ref[k]).max(), 1.0), (k, err, float(ref['spread.' + k]))
    assert _rel(pkg['printed'][:, :4], ref['printed'][:, :4]) <= 2e-3      # what the user sees: the printed losses of every step
