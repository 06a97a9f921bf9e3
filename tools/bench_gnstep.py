import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
import bench
from epropnp.camera import PerspectiveCamera
from epropnp.cost_fun import AdaptiveHuberPnPCost
from epropnp.epropnp import EProPnP6DoF
from epropnp.levenberg_marquardt import LMSolver
from epropnp.losses import monte_carlo_pose_loss
dev = torch.device('cuda:0')
B, N = 4096, 512
p = bench.synth_problem(B, N, dev, seed=5)
x3d, x2d, w2d = (p[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
cam = PerspectiveCamera(cam_mats=p['cam_mats'])
cf = AdaptiveHuberPnPCost(relative_delta=0.5)
layer = EProPnP6DoF(mc_samples=512, num_iter=4, solver=LMSolver(dof=6, num_iter=3))
def step(plus):
    for tt in (x3d, x2d, w2d):
        tt.grad = None
    cf.set_param(x2d.detach(), w2d)
    o = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], force_init_solve=False, with_pose_opt_plus=plus)
    loss = monte_carlo_pose_loss(o[4], o[5]).mean()
    if plus:
        loss = loss + 0.1 * (o[2][:, :3] - p['pose_init'][:, :3]).norm(dim=-1).mean() + 0.1 * (1 - (o[2][:, 3:] * p['pose_init'][:, 3:]).sum(-1).square()).mean()
    loss.backward()
for plus in (False, True):
    for _ in range(3): step(plus)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(plus)
    torch.cuda.synchronize()
    print('with_pose_opt_plus', plus, 'ms/step', round((time.perf_counter() - t0) / 10 * 1e3, 3), 'peak MB', torch.cuda.max_memory_allocated() // 2**20)
