#!/usr/bin/env python
"""End-to-end demo: fit out_pose = EProPnP(MLP(in_pose)) to the identity map on SE(3).

Script form of the reference's demo/fit_identity.ipynb (cells 5-12) on the MI355X package: the same model (MLP ->
64 2D-3D correspondences -> EProPnP6DoF with LMSolver(10) + RSLMSolver(8 pts, 128 proposals, 5 iters)), the same
Monte-Carlo pose loss + derivative regularisation, Adam.  Run on a HIP device:

    python demo/fit_identity.py [--iters 300] [--batch 256]
"""
import argparse
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'epro-pnp_amd'))
from epropnp.camera import PerspectiveCamera  # noqa: E402
from epropnp.cost_fun import AdaptiveHuberPnPCost  # noqa: E402
from epropnp.epropnp import EProPnP6DoF  # noqa: E402
from epropnp.levenberg_marquardt import LMSolver, RSLMSolver  # noqa: E402
from epropnp.losses import MonteCarloPoseLoss  # noqa: E402


class Model(nn.Module):
    def __init__(self, num_points=64, hidden=1024):
        super().__init__()
        self.num_points = num_points
        self.mlp = nn.Sequential(nn.Linear(7, hidden), nn.LeakyReLU(), nn.Linear(hidden, num_points * 7))
        self.log_weight_scale = nn.Parameter(torch.zeros(2))
        self.epropnp = EProPnP6DoF(mc_samples=512, num_iter=4,
                                   solver=LMSolver(dof=6, num_iter=10,
                                                   init_solver=RSLMSolver(dof=6, num_points=8, num_proposals=128, num_iter=5)))
        self.camera = PerspectiveCamera()
        self.cost_fun = AdaptiveHuberPnPCost(relative_delta=0.5)

    def correspondences(self, in_pose):
        x3d, x2d, w2d = self.mlp(in_pose).reshape(-1, self.num_points, 7).split([3, 2, 2], dim=-1)
        w2d = (w2d.log_softmax(dim=-2) + self.log_weight_scale).exp()
        return x3d, x2d, w2d

    def forward_train(self, in_pose, cam_mats, out_pose):
        x3d, x2d, w2d = self.correspondences(in_pose)
        self.camera.set_param(cam_mats)
        self.cost_fun.set_param(x2d.detach(), w2d)
        out = self.epropnp.monte_carlo_forward(x3d, x2d, w2d, self.camera, self.cost_fun, pose_init=out_pose,
                                               force_init_solve=True, with_pose_opt_plus=True)
        return out, self.log_weight_scale.detach().exp().mean()

    def forward_test(self, in_pose, cam_mats):
        x3d, x2d, w2d = self.correspondences(in_pose)
        self.camera.set_param(cam_mats)
        self.cost_fun.set_param(x2d.detach(), w2d)
        return self.epropnp(x3d, x2d, w2d, self.camera, self.cost_fun)[0]


def sample_poses(n, device, gen):
    pose = torch.randn(n, 7, device=device, generator=gen)
    pose[:, 2] += 5
    pose[:, 3:] = F.normalize(pose[:, 3:], dim=-1)
    return pose


def evaluate(model, device, gen, n=1024):
    test = sample_poses(n, device, gen)
    with torch.no_grad():
        pose_opt = model.forward_test(test, torch.eye(3, device=device).expand(n, -1, -1))
    dist_t = (pose_opt[:, :3] - test[:, :3]).norm(dim=-1).mean().item()
    dot = (pose_opt[:, 3:] * test[:, 3:]).sum(-1).abs().clamp(max=1.0)
    return dist_t, (2 * torch.acos(dot)).mean().item()


def train(iters=300, batch=256, noise=0.01, seed=0, device=None, log_every=50, verbose=True):
    device = device or torch.device('cuda:0')
    torch.manual_seed(seed)
    gen = torch.Generator(device=device).manual_seed(seed)
    model = Model().to(device)
    loss_fun = MonteCarloPoseLoss(momentum=0.1).to(device)
    opt = torch.optim.Adam([{'params': model.mlp.parameters()}, {'params': model.log_weight_scale, 'lr': 1e-2}], lr=1e-4)
    cam = torch.eye(3, device=device)
    before = evaluate(model, device, gen)
    for it in range(iters):
        in_pose = sample_poses(batch, device, gen)
        out_pose = in_pose + torch.randn(batch, 7, device=device, generator=gen) * noise
        out_pose[:, 3:] = F.normalize(out_pose[:, 3:], dim=-1)
        (_, _, pose_opt_plus, _, logw, cost_tgt), norm_factor = model.forward_train(in_pose, cam.expand(batch, -1, -1), out_pose)
        loss_mc = loss_fun(logw, cost_tgt, norm_factor)
        dist_t = (pose_opt_plus[:, :3] - out_pose[:, :3]).norm(dim=-1)
        loss_t = torch.where(dist_t < 1.0, 0.5 * dist_t.square(), dist_t - 0.5).mean()
        dot = (pose_opt_plus[:, 3:] * out_pose[:, 3:]).sum(-1)
        loss_r = ((1 - dot.square()) * 2).mean()
        loss = loss_mc + 0.1 * loss_t + 0.1 * loss_r
        opt.zero_grad()
        loss.backward()
        grads = [p.grad.norm() for p in model.parameters() if p.grad is not None]
        if not torch.isfinite(torch.stack(grads)).all():      # NaN-safe step, as the reference's training loops do
            continue
        opt.step()
        if verbose and (it % log_every == 0 or it == iters - 1):
            print(f'iter {it:4d}  loss_mc={loss_mc.item():.4f} loss_t={loss_t.item():.4f} loss_r={loss_r.item():.4f} '
                  f'norm_factor={loss_fun.norm_factor.item():.4f}', flush=True)
    after = evaluate(model, device, gen)
    if verbose:
        print(f'mean translation error {before[0]:.4f} -> {after[0]:.4f}; mean orientation error {before[1]:.4f} -> {after[1]:.4f}')
    return before, after


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=300)
    ap.add_argument('--batch', type=int, default=256)
    a = ap.parse_args()
    train(a.iters, a.batch)
