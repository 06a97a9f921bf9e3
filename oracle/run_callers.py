"""
oracle/run_callers.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (build container only: needs /root/reference).

Executes the LITERAL source of the reference's own callers of the EPro-PnP layer -- read from the reference checkout
at run time, never copied -- against one of two `epropnp` packages:

    --side reference   the unmodified reference (`/root/reference/epropnp`, random draws injected by oracle/ref_runner.py)
    --side package     this repository's package (`epro-pnp_amd/epropnp`) on the CPU emulation build of the kernels
                       (or on cuda:0 when a HIP device is present)

and writes what the caller computed to an .npz.  tests/test_reference_callers.py runs both sides as subprocesses (the two
packages have the same import name, so they cannot share a process) on the same seeded inputs and the same injected random
draws and compares: a missing attribute, keyword, return arity or tensor convention in the package fails the exec, a
numerical disagreement fails the comparison.  This is the "drop it in unchanged" proof of BASELINE.json's north_star.

Scenarios (reference file:line of the executed slices):
  notebook   demo/fit_identity.ipynb code cells 5-10 (imports, Model with EProPnP6DoF + LMSolver + RSLMSolver,
             MonteCarloPoseLoss, data, 3 training steps incl. optimizer), then Model.forward_test of cell 7
  train6dof  EPro-PnP-6DoF/lib/train.py:47-57 (layer construction) and :141-193 (dense correspondences, camera with
             tensor bounds, AdaptiveHuberPnPCost(0.1), monte_carlo_forward, MC loss, derivative regularisation)
  det        EPro-PnP-Det/epropnp_det/models/dense_heads/deform_pnp_head.py:870-893 (pose loss over the stages + the
             pose_opt_plus call) and :514-527 (test_post: plain solve and monte_carlo_forward(fast_mode=True))
Only run-size constants (batch size, number of steps) are overridden after the cell that sets them; no call site is edited.
"""
import argparse
import contextlib
import importlib.util
import io
import json
import math
import os
import re
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_ROOT = os.environ.get('EPROPNP_REFERENCE', '/root/reference')
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def source_lines(path, first, last, must_contain):
    with open(path) as f:
        text = textwrap.dedent(''.join(f.readlines()[first - 1:last]))
    for needle in must_contain:
        assert needle in text, f'{path}:{first}-{last} no longer contains {needle!r}'
    return text


def import_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------------------------------------------------
# the two sides: same interface -- names(), inject(call_index, B, N, layer), loss modules
# ----------------------------------------------------------------------------------------------------------------------
class Draws:
    """Seeded random draws for call number `c` of a scenario: AMIS base noise and RSLM sub-samples / rotations."""

    def __init__(self, seed):
        self.seed, self.calls = seed, 0

    def next(self, B, N, dof, S, K, n, P):
        import epropnp_oracle as orc
        c = self.calls
        self.calls += 1
        noise = orc.make_noise(B, S, K, dof, seed=self.seed + 2 * c)
        g = torch.Generator().manual_seed(self.seed + 2 * c + 1)
        inds = torch.stack([torch.randperm(N, generator=g)[:n] for _ in range(P * B)]).reshape(P, B, n)
        if dof == 4:
            rot = torch.rand(P, B, generator=g) * (2 * math.pi)
        else:
            rot = torch.nn.functional.normalize(torch.randn(P, B, 4, generator=g), dim=-1)
        return noise, dict(inds=inds, rot=rot)


def setup_reference(draws):
    import ref_runner as ref
    m = ref.load_reference()
    Base, LM = m['epropnp'].EProPnPBase, m['levenberg_marquardt'].LMSolver

    def arm(layer, x2d):
        B, N = x2d.shape[:2]
        sv = layer.solver if hasattr(layer, 'solver') else layer
        init = getattr(sv, 'init_solver', None)
        n, P = (init.num_points, init.num_proposals) if init is not None else (1, 1)
        S, K = (layer.mc_samples, layer.num_iter) if hasattr(layer, 'mc_samples') else (4, 1)
        noise, rslm = draws.next(B, N, sv.dof, S, K, min(n, N), P)
        ref.INJ.reset(noise, rslm)
    mc0, fw0 = Base.monte_carlo_forward, Base.forward

    def mc(self, x3d, x2d, *a, **k):
        arm(self, x2d)
        return mc0(self, x3d, x2d, *a, **k)

    def fw(self, x3d, x2d, *a, **k):
        arm(self, x2d)
        return fw0(self, x3d, x2d, *a, **k)
    Base.monte_carlo_forward, Base.forward = mc, fw
    import mmdet_shim
    mmdet_shim.install()
    names = dict(EProPnP6DoF=m['epropnp'].EProPnP6DoF, EProPnP4DoF=m['epropnp'].EProPnP4DoF, LMSolver=LM,
                 RSLMSolver=m['levenberg_marquardt'].RSLMSolver, PerspectiveCamera=m['camera'].PerspectiveCamera,
                 AdaptiveHuberPnPCost=m['cost_fun'].AdaptiveHuberPnPCost, evaluate_pnp=m['common'].evaluate_pnp)
    names['Loss6DoF'] = import_file('ref_loss_6dof', os.path.join(REF_ROOT, 'EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py')).MonteCarloPoseLoss
    names['LossDet'] = import_file('ref_loss_det', os.path.join(REF_ROOT, 'EPro-PnP-Det/epropnp_det/models/losses/monte_carlo_pose_loss.py')).MonteCarloPoseLoss

    def build_det_pnp():        # the Det copy of the layer needs mmcv; the top-level reference package is the same arithmetic
        return names['EProPnP4DoF'](mc_samples=512, num_iter=4, normalize=True, solver=LM(
            dof=4, num_iter=10, normalize=True, init_solver=names['RSLMSolver'](dof=4, num_points=16, num_proposals=64, num_iter=3)))
    names['build_det_pnp'] = build_det_pnp
    return names, torch.device('cpu')


def patch_package(draws):
    """The package's classes under the names the scenarios use, with the layer's entry points wrapped so that every call
    consumes the next seeded draws (AMIS noise through the `noise=` keyword, RSLM sub-samples / rotations through the
    initialiser's `draw` hook).  Returns (names, restore): `restore()` puts the unwrapped methods back (tests)."""
    from epropnp import builder, camera, common, cost_fun, epropnp, levenberg_marquardt, losses
    assert epropnp.__file__.startswith(ROOT)
    from helpers import pack_noise
    Base = epropnp.EProPnPBase

    def arm(layer, x2d, kw):
        B, N = x2d.shape[:2]
        sv = layer.solver if hasattr(layer, 'solver') else layer
        init = getattr(sv, 'init_solver', None)
        n, P = (init.num_points, init.num_proposals) if init is not None else (1, 1)
        S, K = (layer.mc_samples, layer.num_iter) if hasattr(layer, 'mc_samples') else (4, 1)
        noise, rslm = draws.next(B, N, sv.dof, S, K, min(n, N), P)
        if init is not None:      # the package's reproducibility hook: an overridden `draw` injects the sub-samples / rotations
            init.draw = lambda w2d: (rslm['inds'].to(w2d.device), rslm['rot'].to(w2d.device))
        if kw is not None:
            kw['noise'] = pack_noise(noise, sv.dof).to(x2d.device)
    mc0, fw0 = Base.monte_carlo_forward, Base.forward

    def mc(self, x3d, x2d, *a, **k):
        if k.get('noise') is None:           # (the layer re-enters itself for pose_init / cam_mats gradients with the noise set)
            arm(self, x2d, k)
        return mc0(self, x3d, x2d, *a, **k)

    def fw(self, x3d, x2d, *a, **k):
        arm(self, x2d, None)
        return fw0(self, x3d, x2d, *a, **k)
    Base.monte_carlo_forward, Base.forward = mc, fw

    def restore():
        Base.monte_carlo_forward, Base.forward = mc0, fw0
    names = dict(EProPnP6DoF=epropnp.EProPnP6DoF, EProPnP4DoF=epropnp.EProPnP4DoF, LMSolver=levenberg_marquardt.LMSolver,
                 RSLMSolver=levenberg_marquardt.RSLMSolver, PerspectiveCamera=camera.PerspectiveCamera,
                 AdaptiveHuberPnPCost=cost_fun.AdaptiveHuberPnPCost, evaluate_pnp=common.evaluate_pnp,
                 Loss6DoF=losses.MonteCarloPoseLoss, LossDet=losses.MonteCarloPoseLoss)
    # the detection head builds its layer from the config dict (configs/epropnp_det_v1b_220312.py:98-111)
    names['build_det_pnp'] = lambda: builder.build_pnp(dict(
        type='EProPnP4DoF', mc_samples=512, num_iter=4, normalize=True,
        solver=dict(type='LMSolver', num_iter=10, normalize=True,
                    init_solver=dict(type='RSLMSolver', num_points=16, num_proposals=64, num_iter=3))))
    return names, restore


def setup_package(draws):
    sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
    for p in (os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'emu')):
        sys.path.insert(0, p)
    if torch.cuda.is_available():
        dev = torch.device('cuda:0')
    else:
        import conftest
        import install as emu
        emu.install(conftest._emu_lib())
        dev = torch.device('cpu')
    return patch_package(draws)[0], dev


SEEDS = {'notebook': 1000, 'train6dof': 2000, 'det': 3000}
FIXTURES = {'notebook': 'callers_notebook.npz', 'train6dof': 'callers_linemod_train.npz', 'det': 'callers_det_head.npz'}
FIXTURE_RUN = {'notebook': dict(objects=4, steps=3), 'train6dof': dict(objects=4, steps=2), 'det': dict(objects=4, steps=2)}


def write_fixture(scenario, out_path, tmp_dir):
    """tests/golden/callers_*.npz: what the UNMODIFIED reference computes when its own caller's literal source is executed
    (this script, --side reference, in a subprocess: the reference and the package share the import name `epropnp`), stored
    as float32 -- the GPU box compares the package against these (tests/test_callers_gpu.py).  The notebook fixture also
    carries `spread.<key>`: how far each output of the reference itself moves under a 2e-7 relative jitter of the network
    inputs (the yardstick of tests/test_reference_callers.py::test_notebook_cells_run_unchanged)."""
    import subprocess
    run = FIXTURE_RUN[scenario]

    def ref(tag, *extra):
        path = os.path.join(tmp_dir, f'{scenario}_{tag}.npz')
        cmd = [sys.executable, os.path.abspath(__file__), '--side', 'reference', '--scenario', scenario, '--out', path,
               '--objects', str(run['objects']), '--steps', str(run['steps'])] + list(extra)
        subprocess.run(cmd, check=True, capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''))
        return dict(np.load(path))
    res = ref('ref')
    arrays = {k: (v.astype(np.float64) if k == 'printed' else v.astype(np.float32)) for k, v in res.items()}
    if scenario == 'notebook':
        jit = ref('jit', '--jitter', '2e-7')
        for k in res:
            arrays['spread.' + k] = np.float64(np.abs(jit[k] - res[k]).max())
    arrays['meta.objects'], arrays['meta.steps'] = np.int64(run['objects']), np.int64(run['steps'])
    np.savez_compressed(out_path, **arrays)
    return arrays


# ----------------------------------------------------------------------------------------------------------------------
def scenario_notebook(names, dev, bs, steps, jitter=0.0):
    nb = json.load(open(os.path.join(REF_ROOT, 'demo/fit_identity.ipynb')))
    cells = {i: ''.join(c['source']) for i, c in enumerate(nb['cells']) if c['cell_type'] == 'code'}
    for i, needle in ((5, 'from epropnp.epropnp import EProPnP6DoF'), (7, 'self.epropnp.monte_carlo_forward('),
                      (8, 'class MonteCarloPoseLoss'), (9, 'model = Model().to(device)'), (10, 'loss.backward()')):
        assert needle in cells[i], f'notebook cell {i} no longer contains {needle!r}'
    ns = {'__name__': 'notebook'}
    torch.manual_seed(0)
    exec(compile(cells[5], 'fit_identity.ipynb:cell5', 'exec'), ns)          # the imports resolve to the side's `epropnp`
    exec(compile(cells[6], 'fit_identity.ipynb:cell6', 'exec'), ns)
    ns.update(device=dev, n_data=bs * steps, batch_size=bs, n_epoch=1)       # run-size constants only
    for i in (7, 8, 9):
        exec(compile(cells[i], f'fit_identity.ipynb:cell{i}', 'exec'), ns)
    if jitter:       # yardstick run: the network inputs moved by ~1 ulp (how far the reference itself moves under rounding noise)
        g = torch.Generator().manual_seed(77)
        ns['in_pose'].mul_(1 + jitter * torch.randn(ns['in_pose'].shape, generator=g).to(dev))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        exec(compile(cells[10], 'fit_identity.ipynb:cell10', 'exec'), ns)
    rows = [[float(v) for v in re.findall(r'=(-?[0-9.]+(?:e-?\d+)?|nan|inf)', ln)] for ln in buf.getvalue().splitlines()
            if ln.startswith('Epoch')]
    assert len(rows) == steps and all(len(r) == 6 for r in rows), buf.getvalue()
    out = dict(printed=torch.tensor(rows, dtype=torch.float64))             # loss_mc, loss_t, loss_r, loss, norm_factor, grad_norm
    for k in ('pose_opt_plus', 'pose_sample_logweights', 'cost_tgt', 'loss', 'grad_norm'):
        out['last.' + k] = ns[k].detach().double().cpu()
    out['last.norm_factor_buffer'] = ns['mc_loss_fun'].norm_factor.detach().double().cpu()
    with torch.no_grad():                                                    # Model.forward_test of cell 7 (inference path)
        g = torch.Generator().manual_seed(5)
        test_in = torch.randn(bs, 7, generator=g)
        test_in[:, 2] += 5
        test_in[:, 3:] = torch.nn.functional.normalize(test_in[:, 3:], dim=-1)
        if jitter:
            test_in = test_in * (1 + jitter * torch.randn(test_in.shape, generator=g))
        test_in = test_in.to(dev)
        cam = ns['cam_mats'].expand(bs, -1, -1)
        out['test.pose_opt'] = ns['model'].forward_test(test_in, cam).double().cpu()
        out['test.pose_opt_fast'] = ns['model'].forward_test(test_in, cam, fast_mode=True).double().cpu()
    return out


def _linemod_like(bs, res, dev, seed):
    """A consistent dense-correspondence scene: crop boxes around the projected objects, per-pixel NOC maps obtained by
    back-projecting every crop pixel onto a smooth depth surface of the object at the ground-truth pose."""
    g = torch.Generator().manual_seed(seed)
    K = torch.tensor([[572.4114, 0., 325.2611], [0., 573.57043, 242.04899], [0., 0., 1.]])
    ang = torch.randn(bs, 3, generator=g) * 0.6
    th = ang.norm(dim=-1, keepdim=True)
    ax = ang / th
    Kx = torch.zeros(bs, 3, 3)
    Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0], Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    R = torch.eye(3) + torch.sin(th)[..., None] * Kx + (1 - torch.cos(th))[..., None] * (Kx @ Kx)
    t = torch.cat((torch.randn(bs, 2, generator=g) * 0.05, 0.8 + 0.2 * torch.rand(bs, 1, generator=g)), -1)
    dim = 0.06 + 0.05 * torch.rand(bs, 3, generator=g)
    centre = (K @ t[..., None]).squeeze(-1)
    c_box = centre[:, :2] / centre[:, 2:]
    s_box = (2.6 * dim.max(dim=-1).values * K[0, 0] / t[:, 2]).clamp(min=48.0)
    s = s_box.to(torch.int64)
    begin = c_box.to(torch.int64) - s[:, None] / 2.
    unit = s.to(torch.float32) / res
    ar = torch.arange(res, dtype=torch.float32)
    yy, xx = torch.meshgrid(ar, ar, indexing='ij')
    u = begin[:, 0, None, None] + xx * unit[:, None, None]
    v = begin[:, 1, None, None] + yy * unit[:, None, None]
    depth = t[:, 2, None, None] + 0.02 * torch.sin(xx / 9.0)[None] * torch.cos(yy / 7.0)[None]
    ray = torch.stack(((u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)), -1)       # (bs,h,w,3)
    x_cam = ray * depth[..., None]
    x_obj = torch.einsum('bij,bhwi->bhwj', R, x_cam - t[:, None, None, :])                                  # R^T (x - t)
    noc = (x_obj / dim[:, None, None, :]).permute(0, 3, 1, 2) + 0.01 * torch.randn(bs, 3, res, res, generator=g)
    logit = torch.randn(bs, 2, res, res, generator=g)
    scale = 3.0 + torch.rand(bs, 2, generator=g)
    pose = torch.cat((R, t[..., None]), -1)                                                                 # (bs,3,4)
    to = lambda x: x.to(dev)
    return dict(K=to(K), noc=to(noc), dim=to(dim), logit=to(logit), scale=to(scale), c_box=to(c_box), s_box=to(s_box),
                pose=to(pose))


def scenario_train6dof(names, dev, bs, steps):
    train_py = os.path.join(REF_ROOT, 'EPro-PnP-6DoF/lib/train.py')
    src_a = source_lines(train_py, 47, 57, ['epropnp = EProPnP6DoF(', 'num_proposals=4', '.cuda(cfg.pytorch.gpu)'])
    src_b = source_lines(train_py, 141, 193, ['x3d = noc * dim[..., None, None]', 'cost_fun.set_param(x2d, w2d)',
                                              'epropnp.monte_carlo_forward(', 'model.monte_carlo_pose_loss(',
                                              'loss_r = loss_r.mean()'])
    rc = import_file('ref_rotation_conversions', os.path.join(REF_ROOT, 'EPro-PnP-6DoF/lib/ops/rotation_conversions.py'))
    if dev.type == 'cpu':          # `.cuda(gpu)` of the literal constructor line on a box without a GPU: stay where we are
        torch.nn.Module.cuda = lambda self, *a, **k: self
    res = 64                                                   # the slice hard-codes a 64 x 64 map and 512 sampled pixels
    cfg = types.SimpleNamespace(pytorch=types.SimpleNamespace(gpu=0), dataiter=types.SimpleNamespace(out_res=res))
    ns = dict(torch=torch, np=np, math=math, cfg=cfg, matrix_to_quaternion=rc.matrix_to_quaternion, **names)
    exec(compile(src_a, 'lib/train.py:47-57', 'exec'), ns)
    loss_mod = names['Loss6DoF'](momentum=0.01).to(dev)
    ns['model'] = types.SimpleNamespace(monte_carlo_pose_loss=loss_mod)
    out = {}
    np.random.seed(11)
    for it in range(steps):
        sc = _linemod_like(bs, res, dev, seed=70 + it)
        noc, logit, scale = (sc[k].clone().requires_grad_(True) for k in ('noc', 'logit', 'scale'))
        ns.update(noc=noc, dim=sc['dim'], w2d=logit, scale=scale, s_box_var=sc['s_box'], c_box_var=sc['c_box'],
                  pose_var=sc['pose'], bs=bs, cam_intrinsic=sc['K'])
        exec(compile(src_b, 'lib/train.py:141-193', 'exec'), ns)
        loss = ns['loss_mc'] + 0.1 * ns['loss_t'] + 0.1 * ns['loss_r']
        loss.backward()
        for k in ('loss_mc', 'loss_t', 'loss_r', 'pose_opt_plus', 'pose_sample_logweights', 'cost_tgt', 'pose_gt'):
            out[f'step{it}.{k}'] = ns[k].detach().double().cpu()
        out[f'step{it}.g_noc'], out[f'step{it}.g_logit'], out[f'step{it}.g_scale'] = (
            t.grad.double().cpu() for t in (noc, logit, scale))
    out['norm_factor_buffer'] = loss_mod.norm_factor.detach().double().cpu()
    return out


def scenario_det(names, dev, n_obj, steps):
    head = os.path.join(REF_ROOT, 'EPro-PnP-Det/epropnp_det/models/dense_heads/deform_pnp_head.py')
    src_loss = source_lines(head, 870, 893, ['norm_factor = (scale * sample_weights[:, None]).sum()', 'self.pnp.monte_carlo_forward(',
                                             'self.loss_pose[stage_id](', 'with_pose_opt_plus=True)'])
    src_test = source_lines(head, 514, 527, ['self.camera.set_param(cam_intrinsic_, img_shape=ori_shapes_)',
                                             'self.pnp.monte_carlo_forward(', 'fast_mode=True)[0]'])
    import epropnp_oracle as orc
    N, stages = 128, 2
    pnp = names['build_det_pnp']()
    self_ = types.SimpleNamespace(pnp=pnp, camera=names['PerspectiveCamera'](), cost_fun=names['AdaptiveHuberPnPCost'](relative_delta=0.5),
                                  loss_pose=[names['LossDet'](loss_weight=0.15, momentum=0.01).to(dev) for _ in range(stages)],
                                  score_type='te', test_cfg=types.SimpleNamespace())
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
        pg = prob['pose_gt']
        ns = dict(torch=torch, self=self_, scale=scale, sample_weights=torch.rand(n_obj, generator=g).to(dev) + 0.5,
                  cam_intrinsic_samples=prob['cam_mats'].to(dev), ori_shape_samples=torch.tensor([[480., 640.]]).expand(n_obj, 2).to(dev),
                  noc_list=noc_list, w2d_list=w2d_list, dim_decoded=dim_decoded, x2d=prob['x2d'].to(dev),
                  bbox_3d_targets=torch.cat((dim_decoded.detach().cpu(), pg[:, :3], pg[:, 3:]), -1).to(dev),
                  num_obj_samples=float(n_obj), num_obj_actual=n_obj, losses={}, noc=noc_list[-1], w2d=w2d_list[-1])
        exec(compile(src_loss, 'deform_pnp_head.py:870-893', 'exec'), ns)
        tgt = ns['bbox_3d_targets']
        total = sum(ns['losses'].values()) + 0.1 * (ns['pose_opt_plus'][:, :3] - tgt[:, 3:6]).norm(dim=-1).mean() \
            + 0.1 * (ns['pose_opt_plus'][:, 3] - tgt[:, 6]).abs().mean()
        total.backward()
        for k, v in ns['losses'].items():
            out[f'step{it}.{k}'] = v.detach().double().cpu()
        for k in ('pose_opt', 'pose_opt_plus', 'norm_factor'):
            out[f'step{it}.{k}'] = ns[k].detach().double().cpu()
        for s_id in range(stages):
            out[f'step{it}.g_noc{s_id}'] = noc_list[s_id].grad.double().cpu()
            out[f'step{it}.g_w2d{s_id}'] = w2d_list[s_id].grad.double().cpu()
        out[f'step{it}.g_scale'] = scale.grad.double().cpu()
    out['norm_factor_buffers'] = torch.stack([m.norm_factor.detach().double().cpu() for m in self_.loss_pose])
    # ---- test_post :514-527, both branches --------------------------------------------------------------------------
    prob = orc.make_problem(n_obj, N, 4, seed=99)
    for tag, ratio in (('plain', 0.0), ('mc', 0.5)):
        self_.test_cfg = types.SimpleNamespace(mc_scoring_ratio=ratio)
        with torch.no_grad():
            ns = dict(torch=torch, self=self_, cam_intrinsic_=prob['cam_mats'].to(dev),
                      ori_shapes_=torch.tensor([[480., 640.]]).expand(n_obj, 2).to(dev), x3d=prob['x3d'].to(dev), x2d=prob['x2d'].to(dev),
                      w2d=prob['w2d'].to(dev), default_timers={'PnP time': contextlib.nullcontext()}, getattr=getattr)
            exec(compile(src_test, 'deform_pnp_head.py:514-527', 'exec'), ns)
        out[f'test_{tag}.pose_opt'] = ns['pose_opt'].double().cpu()
        if ratio > 0:
            out['test_mc.pose_sample_weights'] = ns['pose_sample_weights'].double().cpu()
            out['test_mc.pose_samples'] = ns['pose_samples'].double().cpu()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--side', choices=['reference', 'package'], required=True)
    ap.add_argument('--scenario', choices=['notebook', 'train6dof', 'det'], required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--objects', type=int, default=4)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--jitter', type=float, default=0.0, help='notebook: relative perturbation of the network inputs (yardstick run)')
    ap.add_argument('--restated', action='store_true', help='run oracle/callers_restated.py (our restatement of the same slices: what the '
                                                            'GPU box runs) instead of the literal reference source')
    a = ap.parse_args()
    assert os.path.isdir(os.path.join(REF_ROOT, 'epropnp')), 'reference checkout not found (build container only)'
    torch.set_num_threads(4)
    draws = Draws(seed=SEEDS[a.scenario])
    names, dev = setup_reference(draws) if a.side == 'reference' else setup_package(draws)
    fn = dict(notebook=scenario_notebook, train6dof=scenario_train6dof, det=scenario_det)[a.scenario]
    if a.restated:
        import callers_restated
        fn = callers_restated.SCENARIOS[a.scenario]
    out = fn(names, dev, a.objects, a.steps, a.jitter) if a.scenario == 'notebook' else fn(names, dev, a.objects, a.steps)
    np.savez(a.out, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print(f'{a.side}/{a.scenario}: {len(out)} arrays, {draws.calls} layer calls -> {a.out}')


if __name__ == '__main__':
    main()
