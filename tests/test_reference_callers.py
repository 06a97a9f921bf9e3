"""The reference's OWN callers, executed literally, on the package (BASELINE.json north_star: "drop it in unchanged").

oracle/run_callers.py reads the source of the reference's callers from the reference checkout at run time -- the code
cells of demo/fit_identity.ipynb, EPro-PnP-6DoF/lib/train.py:47-57,141-193 and
EPro-PnP-Det/.../dense_heads/deform_pnp_head.py:870-893,514-527 -- and exec's it once against the unmodified reference and
once against this repository's `epropnp` package (same import name: two subprocesses), on the same seeded inputs and the
same injected random draws.  An attribute, keyword, return arity or tensor convention the callers rely on and the package
lacks fails the exec; the numbers are compared below.  Build container only (no /root/reference on the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('EPROPNP_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'epropnp')),
                                reason='the reference checkout is only present in the build container')


_CACHE = {}


def _run(side, scenario, out, *extra):
    key = (side, scenario) + tuple(extra)
    if key not in _CACHE:
        _CACHE[key] = _run_uncached(side, scenario, out, *extra)
    return _CACHE[key]


def _run_uncached(side, scenario, out, *extra):
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'run_callers.py'), '--side', side, '--scenario', scenario, '--out', out,
           '--objects', '4'] + list(extra)
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')      # both sides on the CPU (emulation build)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, f'{side}/{scenario} failed:\n{r.stderr[-3000:]}'
    return dict(np.load(out))


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _lse(logw):
    m = logw.max(0)
    return m + np.log(np.exp(logw - m).sum(0))


def test_6dof_training_loop_slice_runs_unchanged(tmp_path):
    """lib/train.py:47-57 + :141-193 on a consistent dense-correspondence scene (64 x 64 maps, 512 sampled pixels, tensor
    bounds, z_min 0.01, relative_delta 0.1, RSLM(16,4,3) + LM 5, S = 512): losses, pose_opt_plus and the gradients that
    reach the network outputs."""
    ref = _run('reference', 'train6dof', str(tmp_path / 'r.npz'), '--steps', '2')
    pkg = _run('package', 'train6dof', str(tmp_path / 'p.npz'), '--steps', '2')
    assert set(ref) == set(pkg)
    for it in range(2):
        for k in ('loss_mc', 'loss_t', 'loss_r', 'cost_tgt'):      # (loss_r = 2 (1 - <q, q_gt>^2) ~ 1e-4: an fp32 cancellation)
            assert np.abs(pkg[f'step{it}.{k}'] - ref[f'step{it}.{k}']).max() <= 1e-6 + 2e-4 * np.abs(ref[f'step{it}.{k}']).max(), (it, k)
        assert np.abs(pkg[f'step{it}.pose_opt_plus'] - ref[f'step{it}.pose_opt_plus']).max() <= 2e-4
        assert np.abs(_lse(pkg[f'step{it}.pose_sample_logweights']) - _lse(ref[f'step{it}.pose_sample_logweights'])).max() <= 1e-3
        for k in ('g_noc', 'g_logit', 'g_scale'):
            assert _rel(pkg[f'step{it}.{k}'], ref[f'step{it}.{k}']) <= 2e-3, (it, k)
    assert abs(float(pkg['norm_factor_buffer']) - float(ref['norm_factor_buffer'])) <= 1e-6


def test_detection_head_slices_run_unchanged(tmp_path):
    """deform_pnp_head.py:870-893 (two-stage pose loss with the Det loss module, then the pose_opt_plus call) with the
    layer built from the config dict, and :514-527 (test_post, both branches): EProPnP4DoF, normalize=True, img_shape
    bounds, RSLM(16,64,3) + LM 10."""
    ref = _run('reference', 'det', str(tmp_path / 'r.npz'), '--steps', '2')
    pkg = _run('package', 'det', str(tmp_path / 'p.npz'), '--steps', '2')
    assert set(ref) == set(pkg)
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


def test_notebook_cells_run_unchanged(tmp_path):
    """demo/fit_identity.ipynb cells 5-10 (3 training steps of the untrained MLP, optimizer included) and forward_test.
    An untrained network emits ill-conditioned correspondences: the reference's own outputs move by ~1e-3 when its inputs
    move by 1 ulp, so the yardstick is the reference against itself under a 2e-7 relative jitter of the network inputs."""
    ref = _run('reference', 'notebook', str(tmp_path / 'r.npz'))
    jit = _run('reference', 'notebook', str(tmp_path / 'j.npz'), '--jitter', '2e-7')
    pkg = _run('package', 'notebook', str(tmp_path / 'p.npz'))
    assert set(ref) == set(pkg)
    assert pkg['printed'].shape == (3, 6) and np.isfinite(pkg['printed']).all()
    for k in ref:
        spread = np.abs(jit[k] - ref[k]).max()
        err = np.abs(pkg[k] - ref[k]).max()
        assert err <= 10 * spread + 1e-3 * max(np.abs(ref[k]).max(), 1.0), (k, err, spread)
    # what the user sees: the printed losses of every step
    assert _rel(pkg['printed'][:, :4], ref['printed'][:, :4]) <= 2e-3


@pytest.mark.parametrize('scenario,steps', [('train6dof', '2'), ('det', '2'), ('notebook', '3')])
def test_restated_slices_are_the_literal_slices(tmp_path, scenario, steps):
    """oracle/callers_restated.py is what runs on the GPU box (tests/test_callers_gpu.py) in place of the literal sources, which only
    exist here: on the same package backend and the same draws the two must agree to the last bit (make_golden.py asserts the same
    on the reference itself), and the committed fixtures must be what the reference computes today."""
    extra = () if scenario == 'notebook' else ('--steps', steps)        # (the notebook's default; same cache keys as the tests above)
    lit = _run('package', scenario, str(tmp_path / 'p.npz'), *extra)
    res = _run('package', scenario, str(tmp_path / 'q.npz'), *extra, '--restated')
    assert set(lit) == set(res)
    for k in lit:
        assert np.array_equal(lit[k], res[k], equal_nan=True), (scenario, k, float(np.abs(lit[k] - res[k]).max()))
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import run_callers
    fix = dict(np.load(os.path.join(ROOT, 'tests', 'golden', run_callers.FIXTURES[scenario])))
    ref = _run('reference', scenario, str(tmp_path / 'r.npz'), *extra)
    for k in ref:
        assert np.array_equal(fix[k], ref[k].astype(fix[k].dtype), equal_nan=True), (scenario, k)
