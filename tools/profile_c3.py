"""C3 (LineMOD training shape: 32 objects x 512 points x 512 samples) under rocprofv3: per-kernel times of a tiny batch."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'epro-pnp_amd'))
import bench
from epropnp.camera import PerspectiveCamera
from epropnp.cost_fun import AdaptiveHuberPnPCost
from epropnp.epropnp import EProPnP6DoF
from epropnp.levenberg_marquardt import LMSolver
from epropnp.losses import monte_carlo_pose_loss
dev = torch.device('cuda:0')
B, N = int(os.environ.get('C3_B', 32)), 512
p = bench.synth_problem(B, N, dev, seed=4)
x3d, x2d, w2d = (p[k].requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
lb = torch.tensor([-200.5, -200.5], device=dev).expand(B, 2).contiguous()
ub = torch.tensor([839.5, 679.5], device=dev).expand(B, 2).contiguous()
cam = PerspectiveCamera(cam_mats=p['cam_mats'], lb=lb, ub=ub)
cf = AdaptiveHuberPnPCost(relative_delta=0.1)
layer = EProPnP6DoF(mc_samples=512, num_iter=4, solver=LMSolver(dof=6, num_iter=5))
for it in range(8):
    for tt in (x3d, x2d, w2d):
        tt.grad = None
    cf.set_param(x2d.detach(), w2d)
    o = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], force_init_solve=False)
    monte_carlo_pose_loss(o[4], o[5]).mean().backward()
torch.cuda.synchronize()
