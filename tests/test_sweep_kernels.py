"""Sweep kernels vs the golden fixtures produced by the unmodified reference (oracle/make_golden.py)."""
import pytest
import torch

from helpers import load_golden, make_layer_objects, set_tune


@pytest.mark.parametrize('name', ['eval6', 'eval6_clip', 'eval4_clip'])
def test_normal_equations_match_reference(backend, name):
    from epropnp import functional as F
    g = load_golden(name)
    p, cam, cf = make_layer_objects(g['prob'], backend)
    prob = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, int(g['dof']))
    jtj, jtr, cost = F.normal_equations(prob, g['pose'].to(backend), clip_jac=True)
    scale = g['jtj'].abs().amax(dim=(-1, -2), keepdim=True)
    assert ((jtj.cpu() - g['jtj']).abs() / scale).max() < 2e-5          # fp32 sums of 2N terms, different order
    assert ((jtr.cpu() - g['jtr']).abs() / g['jtr'].abs().amax(-1, keepdim=True).clamp(min=1e-3)).max() < 1e-4
    torch.testing.assert_close(cost.cpu(), g['cost'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['eval6', 'eval6_clip', 'eval4_clip'])
def test_evaluate_cost_matches_reference(backend, name):
    from epropnp import functional as F
    g = load_golden(name)
    p, cam, cf = make_layer_objects(g['prob'], backend)
    prob = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, int(g['dof']))
    costs = F.evaluate_cost(prob, g['poses'].to(backend))
    torch.testing.assert_close(costs.cpu(), g['costs'], rtol=2e-5, atol=1e-5)
    one = F.evaluate_cost(prob, g['pose'].to(backend))
    torch.testing.assert_close(one.cpu(), g['cost'], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize('dof', [4, 6])
def test_normalize_kernels_match_composite(backend, dof):
    """epropnp_center_points / epropnp_shift_poses vs the PyTorch definition of pnp_normalize / pnp_denormalize
    (reference: epropnp/common.py:103-136)."""
    import epropnp_oracle as orc
    from epropnp import functional as F
    from epropnp.common import pnp_denormalize, pnp_normalize, rotate_offset
    B, N, S = 5, 77, 9
    p = orc.make_problem(B, N, dof=dof, seed=3)
    x3d = (p['x3d'] + torch.tensor([0.3, -1.2, 2.0])).to(backend).requires_grad_(True)
    pose = p['pose_init'].to(backend)
    offset, x3d_n, pose_n = pnp_normalize(x3d, pose)
    ref_off = x3d.detach().mean(dim=-2)
    torch.testing.assert_close(offset, ref_off, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(x3d_n.detach(), x3d.detach() - ref_off.unsqueeze(-2), rtol=1e-6, atol=1e-6)
    ref_pose = torch.cat((pose[..., :3] + rotate_offset(pose, ref_off), pose[..., 3:]), dim=-1)
    torch.testing.assert_close(pose_n, ref_pose, rtol=1e-6, atol=1e-6)
    x3d_n.square().sum().backward()                      # gradient passes through the centring unchanged
    torch.testing.assert_close(x3d.grad, 2 * x3d_n.detach(), rtol=1e-6, atol=1e-6)
    samples = pose_n.unsqueeze(0).repeat(S, 1, 1) + 0.0
    back = pnp_denormalize(offset, samples)
    torch.testing.assert_close(back, pose.unsqueeze(0).expand(S, -1, -1), rtol=1e-5, atol=1e-5)
    assert F.shift_poses(samples, offset, -1.0).shape == samples.shape


@pytest.mark.parametrize('dof', [4, 6])
def test_shift_poses_is_differentiable_in_the_pose(backend, dof):
    """pnp_denormalize of a pose that carries autograd (pose_opt_plus under normalize=True) runs the fused kernel pair;
    values and pose gradients equal the PyTorch definition (reference: epropnp/common.py:127-136)."""
    from epropnp.common import pnp_denormalize, rotate_offset
    g = torch.Generator().manual_seed(dof)
    B = 6
    pose = torch.randn(B, 7 if dof == 6 else 4, generator=g)
    if dof == 6:
        pose[:, 3:] = torch.nn.functional.normalize(pose[:, 3:], dim=-1) * 1.01      # R(q) is not normalised: keep |q| != 1
    offset = torch.randn(B, 3, generator=g)
    up = torch.randn(B, pose.shape[-1], generator=g)
    a = pose.clone().to(backend).requires_grad_(True)
    out = pnp_denormalize(offset.to(backend), a)
    assert 'ShiftPoses' in type(out.grad_fn).__name__ or 'ShiftPoses' in out.grad_fn.name()      # ctypes / C++ node
    (out * up.to(backend)).sum().backward()
    b = pose.clone().double().requires_grad_(True)
    ref = torch.cat((b[..., :3] - rotate_offset(b, offset.double()), b[..., 3:]), dim=-1)
    (ref * up.double()).sum().backward()
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.grad.cpu().double(), b.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('dof,bounds,B,N', [(6, None, 11, 300), (4, 'tight', 7, 64), (6, 'tight', 9, 512), (6, None, 5, 2),
                                            (4, None, 3, 3), (6, 'tight', 4, 449)])
def test_normal_equations_launch_shapes_agree(backend, monkeypatch, dof, bounds, B, N):
    """normal_equations_kernel launch shapes (one wave per object = what bench.py's single-sweep roofline runs at C2,
    two waves, eight waves with the DPP reduction) against the oracle's J^T J / J^T r / cost."""
    import epropnp_oracle as orc
    from epropnp import functional as F
    prob = orc.make_problem(B, N, dof, seed=B + N, bounds=bounds)
    p, cam, cf = make_layer_objects(prob, backend)
    hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
    res, cost, jac = orc.evaluate(*(prob[k].double() for k in ('x3d', 'x2d', 'w2d', 'pose_init')),
                                  orc.Cam(prob['cam_mats'].double(), 0.1, *(None if prob.get(k) is None else prob[k].double() for k in ('lb', 'ub'))),
                                  prob['delta'].double(), True, True)
    jtj = jac.transpose(-1, -2) @ jac
    jtr = (jac.transpose(-1, -2) @ res.unsqueeze(-1)).squeeze(-1)
    shapes = [None, '2,16']         # (an override the kernels are not instantiated for is ignored, not half-applied)
    for w in (2, 8, 16):
        ppl = 1
        while 64 * w * ppl < N:
            ppl *= 2
        if ppl <= 8:
            shapes.append(f'{w},{ppl}')
    for shape in shapes:
        if shape is None:
            set_tune(monkeypatch)
        else:
            set_tune(monkeypatch, ne_shape=shape)
        a, b, c = (t.cpu().double() for t in F.normal_equations(hp, p['pose_init']))
        scale = jtj.abs().amax(dim=(-1, -2), keepdim=True)
        assert ((a - jtj).abs() / scale).max() < 2e-5, shape
        assert ((b - jtr).abs() / jtr.abs().amax(-1, keepdim=True).clamp(min=1e-3)).max() < 1e-4, shape
        torch.testing.assert_close(c, cost, rtol=1e-5, atol=1e-6)


def test_normal_equations_without_points(backend):
    """num_pts = 0: the sweep has nothing to load (the register-resident variants load unconditionally with the index clamped
    to N - 1, so this shape takes the streaming kernel) and writes zeros."""
    from epropnp import functional as F
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import HuberPnPCost
    B = 5
    z = lambda *s: torch.zeros(*s, device=backend)
    cam = PerspectiveCamera(cam_mats=torch.eye(3, device=backend).expand(B, 3, 3).contiguous())
    cf = HuberPnPCost(delta=1.0)
    hp = F.PnPProblem(z(B, 0, 3), z(B, 0, 2), z(B, 0, 2), cam, cf, 6)
    pose = torch.tensor([0., 0., 5., 1., 0., 0., 0.], device=backend).expand(B, 7).contiguous()
    jtj, jtr, cost = F.normal_equations(hp, pose)
    assert jtj.shape == (B, 6, 6) and jtr.shape == (B, 6) and cost.shape == (B,)
    assert float(jtj.abs().max()) == 0.0 and float(jtr.abs().max()) == 0.0 and float(cost.abs().max()) == 0.0
