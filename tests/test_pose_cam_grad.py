"""Gradients of cost_init w.r.t. pose_init and of cost_init / the log-weights w.r.t. camera.cam_mats
(reference: epropnp/epropnp.py:121-124,139-169 records them through autograd; common.py:30-36 is the rotation form it
differentiates) -- the HIP path against fp64 autograd of the oracle."""
import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects, pack_noise


def _oracle_grads(prob, dof, poses, weights, bounds):
    """fp64 autograd of sum_j weights[j] cost(poses[j]) w.r.t. poses and cam_mats (oracle = the reference's arithmetic)."""
    x3d, x2d, w2d = (prob[k].double() for k in ('x3d', 'x2d', 'w2d'))
    K = prob['cam_mats'].double().clone().requires_grad_(True)
    ps = poses.double().clone().requires_grad_(True)
    cam = orc.Cam(K, float(prob.get('z_min', 0.1)), None if bounds is None else prob['lb'].double(),
                  None if bounds is None else prob['ub'].double())
    cost = orc.evaluate(x3d, x2d, w2d, ps, cam, prob['delta'].double(), want_cost=True)[1]
    (cost * weights.double()).sum().backward()
    return ps.grad, K.grad, cost.detach()


@pytest.mark.parametrize('dof,bounds,N', [(6, None, 150), (6, 'tight', 96), (4, None, 64), (4, 'tight', 130)])
def test_pose_cam_grad_kernel_matches_autograd(backend, dof, bounds, N):
    from epropnp import functional as F
    B, P = 5, 3
    prob = orc.make_problem(B, N, dof, seed=31, bounds=bounds)
    g = torch.Generator().manual_seed(5)
    poses = prob['pose_init'].unsqueeze(0).repeat(P, 1, 1)
    poses[1:, :, :3] += 0.05 * torch.randn(P - 1, B, 3, generator=g)
    weights = torch.randn(P, B, generator=g)
    weights[2, 1] = 0.0                                  # a skipped pose
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    for m in (0, 2):
        gp, gk = F.pose_cam_grad(hp, poses.to(backend), weights.to(backend), m_pose=m)
        ref_p, ref_k, _ = _oracle_grads(prob, dof, poses, weights, bounds)
        for mine, ref in ((gp, ref_p[m]), (gk, ref_k)):
            err = (mine.cpu().double() - ref).abs().reshape(B, -1).amax(1) / ref.abs().reshape(B, -1).amax(1).clamp(min=1e-12)
            assert err.max().item() <= 2e-4, (dof, bounds, m, err)
    assert F.pose_cam_grad(hp, poses.to(backend), None, m_pose=-1, want_cam=True)[0] is None


@pytest.mark.parametrize('dof,normalize', [(6, False), (4, True)])
def test_layer_gradients_reach_pose_init_and_cam_mats(backend, dof, normalize):
    """monte_carlo_forward: d(loss)/d pose_init through cost_init and d(loss)/d cam_mats through cost_init AND the
    log-weights, against autograd of the oracle at the kernel's own samples; gradients to the points are unchanged."""
    from epropnp.camera import PerspectiveCamera
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    B, N, S, K = 4, 96, 64, 4
    prob = orc.make_problem(B, N, dof, seed=37)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=38), dof)
    p, cam0, cf = make_layer_objects(prob, backend)
    cls = EProPnP6DoF if dof == 6 else EProPnP4DoF
    layer = cls(mc_samples=S, num_iter=K, normalize=normalize, solver=LMSolver(dof=dof, num_iter=4))
    g = torch.Generator().manual_seed(7)
    g_logw, g_init = torch.randn(S, B, generator=g) * 0.1, torch.randn(B, generator=g)

    def run(with_extra):
        x3d, x2d, w2d = (p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
        pose_init = p['pose_init'].clone().requires_grad_(with_extra)
        Kmat = p['cam_mats'][0].clone().requires_grad_(with_extra)          # ONE (3,3) matrix expanded over the objects
        cam = PerspectiveCamera(cam_mats=Kmat.expand(B, 3, 3), z_min=0.1)
        out = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=pose_init, force_init_solve=False,
                                        noise=noise.to(backend))
        ((out[4] * g_logw.to(backend)).sum() + (out[5] * g_init.to(backend)).sum()).backward()
        return out, x3d.grad, pose_init.grad, Kmat.grad
    out0, gx0, gp0, gk0 = run(False)
    out1, gx1, gp1, gk1 = run(True)
    assert gp0 is None and gk0 is None
    assert torch.equal(out0[4], out1[4]) and torch.equal(out0[5], out1[5]) and torch.equal(gx0, gx1)
    samples = out1[3].detach().cpu()
    # oracle: cost_init term w.r.t. pose_init; cost of pose_init and of every sample w.r.t. the shared K
    ref_p, _, _ = _oracle_grads(prob, dof, prob['pose_init'].unsqueeze(0), g_init.unsqueeze(0), None)
    allp = torch.cat((samples, prob['pose_init'].unsqueeze(0)), 0)
    allw = torch.cat((-g_logw, g_init.unsqueeze(0)), 0)
    _, ref_k, _ = _oracle_grads(prob, dof, allp, allw, None)
    err_p = (gp1.cpu().double() - ref_p[0]).abs().amax(1) / ref_p[0].abs().amax(1)
    assert err_p.max().item() <= 2e-4, err_p
    ref_ksum = ref_k.sum(0)
    assert ((gk1.cpu().double() - ref_ksum).abs().max() / ref_ksum.abs().max()).item() <= 2e-4


@pytest.mark.parametrize('dof,normalize', [(6, False), (4, True)])
def test_pose_opt_plus_is_differentiable_wrt_cam_mats(backend, dof, normalize):
    """with_pose_opt_plus=True and camera.cam_mats requiring grad: the reference differentiates LMSolver.gn_step + pose_add w.r.t. the
    intrinsics too (levenberg_marquardt.py:70-72,243-265).  The fused Gauss-Newton kernels do not; the layer then takes the
    PyTorch composite for pose_opt_plus -- same value to rounding, gradients to the intrinsics AND to the correspondences against
    fp64 autograd of the oracle's gn_step at the layer's pose_opt (round 4 only warned here)."""
    import warnings
    from epropnp.camera import PerspectiveCamera
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    B, N, S, K = 4, 96, 32, 2
    prob = orc.make_problem(B, N, dof, seed=43)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=44), dof)
    p, _, cf = make_layer_objects(prob, backend)
    cls = EProPnP6DoF if dof == 6 else EProPnP4DoF
    layer = cls(mc_samples=S, num_iter=K, normalize=normalize, solver=LMSolver(dof=dof, num_iter=4))
    g = torch.Generator().manual_seed(9)
    g_plus = torch.randn(B, 7 if dof == 6 else 4, generator=g)

    def run(cam_grad):
        x3d, x2d, w2d = (p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
        Kmat = p['cam_mats'].clone().requires_grad_(cam_grad)
        cam = PerspectiveCamera(cam_mats=Kmat, z_min=0.1)
        with warnings.catch_warnings():
            warnings.simplefilter('error')                  # (the round-4 RuntimeWarning must be gone)
            out = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], force_init_solve=False,
                                            noise=noise.to(backend), with_pose_opt_plus=True)
        (out[2] * g_plus.to(backend)).sum().backward()
        return out, x3d.grad, x2d.grad, w2d.grad, Kmat.grad
    out0, gx0, gu0, gw0, gk0 = run(False)           # fused kernels
    out1, gx1, gu1, gw1, gk1 = run(True)            # composite
    assert gk0 is None and gk1 is not None
    assert (out0[2] - out1[2]).abs().max().item() <= 2e-5 and torch.equal(out0[0], out1[0])
    # fp64 autograd of the oracle at the layer's pose_opt
    x3d, x2d, w2d = (prob[k].double().clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    Kd = prob['cam_mats'].double().clone().requires_grad_(True)
    pose = out1[0].detach().cpu().double()
    x3d_s, pose_s, offset = x3d, pose, None
    if normalize:
        offset, x3d_s, pose_s = orc.pnp_normalize(x3d, pose)
    plus = orc.pose_add(pose_s, orc.gn_step(x3d_s, x2d, w2d, pose_s, orc.Cam(Kd, 0.1, None, None), prob['delta'].double()))
    if normalize:
        plus = orc.pnp_denormalize(offset, plus)
    (plus * g_plus.double()).sum().backward()
    for name, mine, ref in (('cam_mats', gk1, Kd.grad), ('x3d', gx1, x3d.grad), ('x2d', gu1, x2d.grad), ('w2d', gw1, w2d.grad),
                            ('x3d (kernels)', gx0, x3d.grad)):
        err = (mine.cpu().double() - ref).abs().reshape(B, -1).amax(1) / ref.abs().reshape(B, -1).amax(1).clamp(min=1e-12)
        assert err.max().item() <= 2e-3, (name, err)
