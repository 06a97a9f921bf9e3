import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
import bench
from epropnp.camera import PerspectiveCamera
from epropnp.cost_fun import AdaptiveHuberPnPCost
from epropnp.epropnp import EProPnP4DoF
from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
from epropnp.losses import monte_carlo_pose_loss
dev = torch.device('cuda:0')
B, N = 600, 128
p = bench.synth_problem(B, N, dev, seed=5, dof=4)
x3d, x2d, w2d = (p[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
cam = PerspectiveCamera(z_min=0.1, allowed_border=200)
cam.set_param(p['cam_mats'], img_shape=torch.tensor([[480., 640.]], device=dev).expand(B, 2))
cf4 = AdaptiveHuberPnPCost(relative_delta=0.5)
layer4 = EProPnP4DoF(mc_samples=128, num_iter=4, normalize=True,
                     solver=LMSolver(dof=4, num_iter=5, init_solver=RSLMSolver(dof=4, num_points=16, num_proposals=64, num_iter=3)))
for it in range(8):
    for tt in (x3d, x2d, w2d):
        tt.grad = None
    cf4.set_param(x2d.detach(), w2d)
    plus = os.environ.get('C4_PLUS', '0') == '1'
    o = layer4.monte_carlo_forward(x3d, x2d, w2d, cam, cf4, pose_init=p['pose_init'], force_init_solve=True,
                                   with_pose_opt_plus=plus)
    loss = monte_carlo_pose_loss(o[4], o[5]).mean()
    if plus:
        loss = loss + 0.1 * (o[2][:, :3] - p['pose_init'][:, :3]).norm(dim=-1).mean()
    loss.backward()
torch.cuda.synchronize()
try:      # tuning builds (build.py -D PNP_TUNING, EPROPNP_LIB=...): per-phase cycle shares of the forward kernel
    import ctypes
    from epropnp import _hip
    fn = _hip.lib().epropnp_tuning_phase_cycles
    buf = (ctypes.c_ulonglong * 8)()
    tot = float(sum(buf[:6])) if fn(buf, 0) == 0 else 0.0
    if tot > 0:
        print('forward phases [init, draw, sweep, weights, refit, store | refit fp64 part, rotation draw]:', [round(v / tot, 3) for v in buf[:8]],
              'cycles/block', round(tot / (8 * B)))
except AttributeError:
    pass
try:
    fn = _hip.lib().epropnp_tuning_rslm_cycles
    buf = (ctypes.c_ulonglong * 8)()
    tot = float(sum(buf[:5])) if fn(buf, 0) == 0 else 0.0
    if tot > 0:
        print('rslm phases [stage+centre, keys, picks, solve, score]:', [round(v / tot, 3) for v in buf[:5]],
              'cycles/block', round(tot / (8 * B)))
except (AttributeError, NameError):
    pass
