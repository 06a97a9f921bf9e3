"""
oracle/callers_restated.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's three callers of the EPro-PnP layer, RESTATED (statement for statement, in this repository's words, every
block citing the reference lines it follows) so that they can run where the reference checkout is absent -- the GPU box.
oracle/run_callers.py executes the LITERAL sources of the same slices (read from /root/reference at run time) against the
unmodified reference and against the package; this file is what lets the same computation run on cuda:0:

    reference, literal slices (build container)  --make_golden.py-->  tests/golden/callers_*.npz   (committed fixtures)
    package, restated slices on cuda:0 (GPU box)  vs  those fixtures                               (tests/test_callers_gpu.py)
    package, restated slices  ==  package, literal slices on the same backend (build container)    (tests/test_reference_callers.py)

Each scenario has the signature of its literal twin in run_callers.py -- (names, dev, objects, steps) -> {key: tensor} with the
same keys -- and takes the classes to use from `names` (run_callers.package_names / setup_reference), so nothing here imports
the product package by itself.  Random inputs are drawn on the CPU and moved to `dev`: the fixtures were produced on the CPU.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.data as Data


def _smooth_l1(dist, beta):
    return torch.where(dist < beta, 0.5 * dist.square() / beta, dist - 0.5 * beta)


def _quat_alignment(pose_plus, pose_gt):
    """2 (1 - <q, q_gt>^2), as a batched (1x4)(4x1) product (notebook cell 10; lib/train.py:191-193)"""
    dot = (pose_plus[:, None, 3:] @ pose_gt[:, 3:, None]).squeeze(-1).squeeze(-1)
    return (1 - dot.square()) * 2


# ----------------------------------------------------------------------------------------------------------------------
# demo/fit_identity.ipynb, cells 5-10 + Model.forward_test
# ----------------------------------------------------------------------------------------------------------------------
def scenario_notebook(names, dev, bs, steps, jitter=0.0):
    EProPnP6DoF, LMSolver, RSLMSolver = names['EProPnP6DoF'], names['LMSolver'], names['RSLMSolver']
    n_points, noise = 64, 0.01                                              # cell 6 (`noise`), cell 7 (`num_points`)
    torch.manual_seed(0)

    # cell 7: the layer, camera and cost function are DEFAULT ARGUMENTS of Model.__init__, i.e. built before anything random
    layer = EProPnP6DoF(mc_samples=512, num_iter=4,
                        solver=LMSolver(dof=6, num_iter=10, init_solver=RSLMSolver(dof=6, num_points=8, num_proposals=128, num_iter=5)))
    camera, cost_fun = names['PerspectiveCamera'](), names['AdaptiveHuberPnPCost'](relative_delta=0.5)

    # cell 9, in its order of random draws: data first (on the CPU: the fixture's device), then the model
    in_pose = torch.randn([bs * steps, 7])
    in_pose[:, 2] += 5
    in_pose[:, 3:] = F.normalize(in_pose[:, 3:], dim=-1)
    out_pose = in_pose + torch.randn([bs * steps, 7]) * noise
    out_pose[:, 3:] = F.normalize(out_pose[:, 3:], dim=-1)
    in_pose, out_pose = in_pose.to(dev), out_pose.to(dev)
    cam_mats = torch.eye(3, device=dev)
    loader = Data.DataLoader(dataset=Data.TensorDataset(in_pose, out_pose), batch_size=bs, shuffle=True)

    class Model(nn.Module):                                                 # cell 7
        def __init__(self):
            super().__init__()
            self.mlp = nn.Sequential(nn.Linear(7, 1024), nn.LeakyReLU(), nn.Linear(1024, n_points * (3 + 2 + 2)))
            self.log_weight_scale = nn.Parameter(torch.zeros(2))
            self.epropnp, self.camera, self.cost_fun = layer, camera, cost_fun

        def forward_correspondence(self, pose):
            x3d, x2d, w2d = self.mlp(pose).reshape(-1, n_points, 7).split([3, 2, 2], dim=-1)
            return x3d, x2d, (w2d.log_softmax(dim=-2) + self.log_weight_scale).exp()

        def forward_train(self, pose, cams, target):
            x3d, x2d, w2d = self.forward_correspondence(pose)
            self.camera.set_param(cams)
            self.cost_fun.set_param(x2d.detach(), w2d)
            res = self.epropnp.monte_carlo_forward(x3d, x2d, w2d, self.camera, self.cost_fun, pose_init=target,
                                                   force_init_solve=True, with_pose_opt_plus=True)
            return (*res, self.log_weight_scale.detach().exp().mean())

        def forward_test(self, pose, cams, fast_mode=False):
            x3d, x2d, w2d = self.forward_correspondence(pose)
            self.camera.set_param(cams)
            self.cost_fun.set_param(x2d.detach(), w2d)
            return self.epropnp(x3d, x2d, w2d, self.camera, self.cost_fun, fast_mode=fast_mode)[0]

    class NotebookLoss(nn.Module):                                          # cell 8 (the notebook defines its own loss module)
        def __init__(self, init_norm_factor=1.0, momentum=0.1):
            super().__init__()
            self.register_buffer('norm_factor', torch.tensor(init_norm_factor, dtype=torch.float))
            self.momentum = momentum

        def forward(self, logweights, cost_target, norm_factor):
            if self.training:
                with torch.no_grad():
                    self.norm_factor.mul_(1 - self.momentum).add_(self.momentum * norm_factor)
            per_obj = cost_target + torch.logsumexp(logweights, dim=0)
            per_obj[torch.isnan(per_obj)] = 0
            return (per_obj.mean() / self.norm_factor).mean()

    model = Model().to(dev)
    mc_loss_fun = NotebookLoss().to(dev)
    optimizer = torch.optim.Adam([{'params': model.mlp.parameters()}, {'params': model.log_weight_scale, 'lr': 1e-2}], lr=1e-4)
    if jitter:
        g = torch.Generator().manual_seed(77)
        in_pose.mul_(1 + jitter * torch.randn(in_pose.shape, generator=g).to(dev))

    rows, last = [], {}
    for batch_in, batch_out in loader:                                      # cell 10, one epoch
        res = model.forward_train(batch_in, cam_mats.expand(batch_in.size(0), -1, -1), batch_out)
        pose_opt_plus, logw, cost_tgt, norm_factor = res[2], res[4], res[5], res[6]
        loss_mc = mc_loss_fun(logw, cost_tgt, norm_factor)
        loss_t = _smooth_l1((pose_opt_plus[:, :3] - batch_out[:, :3]).norm(dim=-1), 1.0).mean()
        loss_r = _quat_alignment(pose_opt_plus, batch_out).mean()
        loss = loss_mc + 0.1 * loss_t + 0.1 * loss_r
        optimizer.zero_grad()
        loss.backward()
        grad_norm = torch.norm(torch.stack([torch.norm(p.grad.detach()) for p in model.parameters()
                                            if p.grad is not None and p.requires_grad]))
        optimizer.step()
        # the notebook prints these six numbers with 4 decimals, and the literal scenario parses them back from the print-out
        rows.append([float(f'{float(v.detach()):.4f}') for v in (loss_mc, loss_t, loss_r, loss, norm_factor, grad_norm)])
        last = dict(pose_opt_plus=pose_opt_plus, pose_sample_logweights=logw, cost_tgt=cost_tgt, loss=loss, grad_norm=grad_norm)
    assert len(rows) == steps
    out = dict(printed=torch.tensor(rows, dtype=torch.float64))
    for k, v in last.items():
        out['last.' + k] = v.detach().double().cpu()
    out['last.norm_factor_buffer'] = mc_loss_fun.norm_factor.detach().double().cpu()
    with torch.no_grad():                                                   # Model.forward_test (cell 7), both solver modes
        g = torch.Generator().manual_seed(5)
        test_in = torch.randn(bs, 7, generator=g)
        test_in[:, 2] += 5
        test_in[:, 3:] = F.normalize(test_in[:, 3:], dim=-1)
        if jitter:
            test_in = test_in * (1 + jitter * torch.randn(test_in.shape, generator=g))
        test_in = test_in.to(dev)
        cams = cam_mats.expand(bs, -1, -1)
        out['test.pose_opt'] = model.forward_test(test_in, cams).double().cpu()
        out['test.pose_opt_fast'] = model.forward_test(test_in, cams, fast_mode=True).double().cpu()
    return out


# ----------------------------------------------------------------------------------------------------------------------
# EPro-PnP-6DoF/lib/train.py:47-57 (layer) and :141-193 (dense correspondences -> layer -> losses)
# ----------------------------------------------------------------------------------------------------------------------
def _rotmat_to_quat_wxyz(R):
    """EPro-PnP-6DoF/lib/ops/rotation_conversions.py: matrix_to_quaternion (the pytorch3d routine: four candidate
    quaternions from the 'square-rooted' diagonal combinations, the best-conditioned one picked per matrix)."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, -1)
    x = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = torch.zeros_like(x)
    pos = x > 0
    q_abs[pos] = torch.sqrt(x[pos])
    cand = torch.stack([torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
                        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
                        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
                        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].max(q_abs.new_tensor(0.1)))
    pick = F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    return cand[pick, :].reshape(R.shape[:-2] + (4,))


def scenario_train6dof(names, dev, bs, steps, scene=None, to_quat=None):
    from run_callers import _linemod_like
    scene = scene or _linemod_like
    to_quat = to_quat or _rotmat_to_quat_wxyz
    res = 64
    # lib/train.py:47-57
    layer = names['EProPnP6DoF'](mc_samples=512, num_iter=4,
                                 solver=names['LMSolver'](dof=6, num_iter=5, init_solver=names['RSLMSolver'](
                                     dof=6, num_points=16, num_proposals=4, num_iter=3)))
    loss_mod = names['Loss6DoF'](momentum=0.01).to(dev)
    out = {}
    np.random.seed(11)
    for it in range(steps):
        sc = scene(bs, res, dev, seed=70 + it)
        noc, logit, scale = (sc[k].clone().requires_grad_(True) for k in ('noc', 'logit', 'scale'))
        # :141-151 object coordinates, pixel grid of the crop
        x3d = noc * sc['dim'][..., None, None]
        side = sc['s_box'].to(torch.int64)
        begin = sc['c_box'].to(torch.int64) - side[:, None] / 2.
        unit = side.to(torch.float32) / res
        ar = torch.arange(res, device=dev, dtype=torch.float32)
        yy, xx = torch.meshgrid(ar, ar, indexing='ij')
        x2d = torch.stack((begin[:, 0, None, None] + xx * unit[:, None, None], begin[:, 1, None, None] + yy * unit[:, None, None]), dim=1)
        # :152-155 ground-truth pose as [t, q]
        pose_gt = torch.cat((sc['pose'][:, :, 3], to_quat(sc['pose'][:, :, :3])), dim=-1)
        # :157-165 512 random pixels per crop; the mean-normalised exponential instead of a softmax
        picks = [np.random.choice(64 * 64, size=64 * 64 // 8, replace=False) for _ in range(bs)]
        inds = x2d.new_tensor(np.asarray(picks), dtype=torch.int64)
        rows = torch.arange(bs, device=dev)[:, None]
        x3d, x2d, w2d = (t.flatten(2).transpose(-1, -2)[rows, inds] for t in (x3d, x2d, logit))
        w2d = (w2d - w2d.mean(dim=1, keepdim=True) - math.log(w2d.size(1))).exp() * scale[:, None, :]
        # :169-180 camera with the crop box -/+ 30 output pixels as projection bounds, adaptive Huber threshold, the layer
        border = 30 * unit
        camera = names['PerspectiveCamera'](cam_mats=sc['K'][None].expand(bs, -1, -1), z_min=0.01, lb=begin - border[:, None],
                                            ub=begin + (res - 1) * unit[:, None] + border[:, None])
        cost_fun = names['AdaptiveHuberPnPCost'](relative_delta=0.1)
        cost_fun.set_param(x2d, w2d)
        _, _, pose_opt_plus, _, logw, cost_tgt = layer.monte_carlo_forward(x3d, x2d, w2d, camera, cost_fun, pose_init=pose_gt,
                                                                          force_init_solve=True, with_pose_opt_plus=True)
        # :182-193 losses
        loss_mc = loss_mod(logw, cost_tgt, scale.detach().mean())
        loss_t = _smooth_l1((pose_opt_plus[:, :3] - pose_gt[:, :3]).norm(dim=-1), 0.05).mean()
        loss_r = _quat_alignment(pose_opt_plus, pose_gt).mean()
        (loss_mc + 0.1 * loss_t + 0.1 * loss_r).backward()
        for k, v in dict(loss_mc=loss_mc, loss_t=loss_t, loss_r=loss_r, pose_opt_plus=pose_opt_plus, pose_sample_logweights=logw,
                         cost_tgt=cost_tgt, pose_gt=pose_gt).items():
            out[f'step{it}.{k}'] = v.detach().double().cpu()
        out[f'step{it}.g_noc'], out[f'step{it}.g_logit'], out[f'step{it}.g_scale'] = (t.grad.double().cpu() for t in (noc, logit, scale))
    out['norm_factor_buffer'] = loss_mod.norm_factor.detach().double().cpu()
    return out


# ----------------------------------------------------------------------------------------------------------------------
# EPro-PnP-Det/epropnp_det/models/dense_heads/deform_pnp_head.py:870-893 (pose loss per stage, pose_opt_plus) and :514-527
# ----------------------------------------------------------------------------------------------------------------------
def scenario_det(names, dev, n_obj, steps):
    import epropnp_oracle as orc
    N, stages = 128, 2
    pnp = names['build_det_pnp']()
    camera, cost_fun = names['PerspectiveCamera'](), names['AdaptiveHuberPnPCost'](relative_delta=0.5)
    loss_pose = [names['LossDet'](loss_weight=0.15, momentum=0.01).to(dev) for _ in range(stages)]
    out = {}
    for it in range(steps):
        prob = orc.make_problem(n_obj, N, 4, seed=90 + it)
        g = torch.Generator().manual_seed(190 + it)
        dim_decoded = (1.0 + torch.rand(n_obj, 3, generator=g)).to(dev)
        x3d0 = prob['x3d'].to(dev)
        noc_list = [((x3d0 + 0.02 * k * torch.randn(n_obj, N, 3, generator=g).to(dev)) / dim_decoded[:, None]).requires_grad_(True)
                    for k in range(stages)]
        w2d_list = [torch.softmax(torch.randn(n_obj, N, 2, generator=g), dim=1).to(dev).requires_grad_(True) for _ in range(stages)]
        scale = (1.5 + torch.rand(n_obj, 2, generator=g)).to(dev).requires_grad_(True)
        sample_weights = torch.rand(n_obj, generator=g).to(dev) + 0.5
        x2d, pg = prob['x2d'].to(dev), prob['pose_gt']
        targets = torch.cat((dim_decoded.detach().cpu(), pg[:, :3], pg[:, 3:]), -1).to(dev)
        # :870-883 the running normaliser's input, then one Monte-Carlo pose loss per decoder stage
        norm_factor = (scale * sample_weights[:, None]).sum() / max(scale.size(0) * 2, 1)
        camera.set_param(prob['cam_mats'].to(dev), img_shape=torch.tensor([[480., 640.]]).expand(n_obj, 2).to(dev))
        losses = {}
        for stage_id, (noc, w2d) in enumerate(zip(noc_list, w2d_list)):
            x3d = noc * dim_decoded[:, None]
            w2d_scaled = w2d * scale[:, None, :]
            cost_fun.set_param(x2d.detach(), w2d_scaled)
            res = pnp.monte_carlo_forward(x3d, x2d, w2d_scaled, camera, cost_fun, pose_init=targets[:, 3:], force_init_solve=True)
            losses[f'loss_pose_{stage_id}'] = loss_pose[stage_id](res[4], res[5], norm_factor, weight=sample_weights,
                                                                  avg_factor=float(n_obj))
        # :885-893 the last stage once more through forward(): pose_opt_plus with dim / scale detached, delta detached
        cost_fun.delta = cost_fun.delta.detach()
        pose_opt, _, _, pose_opt_plus = pnp(noc * dim_decoded[:, None].detach(), x2d, w2d * scale[:, None, :].detach(), camera, cost_fun,
                                            with_pose_opt_plus=True)
        total = sum(losses.values()) + 0.1 * (pose_opt_plus[:, :3] - targets[:, 3:6]).norm(dim=-1).mean() \
            + 0.1 * (pose_opt_plus[:, 3] - targets[:, 6]).abs().mean()
        total.backward()
        for k, v in losses.items():
            out[f'step{it}.{k}'] = v.detach().double().cpu()
        for k, v in dict(pose_opt=pose_opt, pose_opt_plus=pose_opt_plus, norm_factor=norm_factor).items():
            out[f'step{it}.{k}'] = v.detach().double().cpu()
        for s_id in range(stages):
            out[f'step{it}.g_noc{s_id}'] = noc_list[s_id].grad.double().cpu()
            out[f'step{it}.g_w2d{s_id}'] = w2d_list[s_id].grad.double().cpu()
        out[f'step{it}.g_scale'] = scale.grad.double().cpu()
    out['norm_factor_buffers'] = torch.stack([m.norm_factor.detach().double().cpu() for m in loss_pose])
    # :514-527 test_post: fast-mode solve, or the sampler in fast mode with softmax-ed weights
    prob = orc.make_problem(n_obj, N, 4, seed=99)
    x3d, x2d, w2d = (prob[k].to(dev) for k in ('x3d', 'x2d', 'w2d'))
    for tag, ratio in (('plain', 0.0), ('mc', 0.5)):
        with torch.no_grad():
            camera.set_param(prob['cam_mats'].to(dev), img_shape=torch.tensor([[480., 640.]]).expand(n_obj, 2).to(dev))
            cost_fun.set_param(x2d.detach(), w2d)
            if ratio > 0:
                pose_opt, _, _, samples, logw, _ = pnp.monte_carlo_forward(x3d, x2d, w2d, camera, cost_fun, fast_mode=True)
                out['test_mc.pose_sample_weights'] = logw.softmax(dim=0).double().cpu()
                out['test_mc.pose_samples'] = samples.double().cpu()
            else:
                pose_opt = pnp(x3d, x2d, w2d, camera, cost_fun, fast_mode=True)[0]
        out[f'test_{tag}.pose_opt'] = pose_opt.double().cpu()
    return out


SCENARIOS = dict(notebook=scenario_notebook, train6dof=scenario_train6dof, det=scenario_det)
