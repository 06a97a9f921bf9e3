import os, sys, torch
sys.path[:0] = ['/root/repo/epro-pnp_amd', '/root/repo/oracle', '/root/repo/tests']
import epropnp_oracle as orc
from helpers import make_layer_objects
from epropnp import functional as F
dev = torch.device('cuda:0')
B, N, S, dof = int(sys.argv[1]), int(sys.argv[2]), 96, 6
bounded = len(sys.argv) > 3 and sys.argv[3] == '1'
prob = orc.make_problem(B, N, dof, seed=41, relative_delta=0.1)
if bounded:
    lo, hi = prob['x2d'].amin(1), prob['x2d'].amax(1)
    unit = (hi - lo).amax(-1, keepdim=True) / 64.0
    prob['lb'], prob['ub'], prob['z_min'] = (lo - 30 * unit).contiguous(), (hi + 30 * unit).contiguous(), 0.01
g = torch.Generator().manual_seed(7)
poses = prob['pose_gt'].unsqueeze(0).repeat(S, 1, 1)
poses[..., :3] += 0.2 * torch.randn(S, B, 3, generator=g)
q = poses[..., 3:] + 0.1 * torch.randn(S, B, 4, generator=g)
poses[..., 3:] = q / q.norm(dim=-1, keepdim=True)
g_logw, g_init = torch.randn(S, B, generator=g), torch.randn(B, generator=g)
p, cam, cf = make_layer_objects(prob, dev)
hp = F.PnPProblem(p['x3d'], p['x2d'], p['w2d'], cam, cf, dof)
args = (hp, poses.to(dev), g_logw.to(dev), p['pose_init'], g_init.to(dev))
runs = []
for rep in range(6):
    o = F.amis_backward(*args, nsplit=1)
    torch.cuda.synchronize()
    runs.append([t.clone() for t in o[:3]])
# majority reference per element: median over runs is robust if corruption is rare
stack = torch.stack([r[0] for r in runs])
ref = stack.median(0).values
summary = []
for i, r in enumerate(runs):
    d = (r[0] - ref).abs().amax(-1)
    bad = (d > 0).nonzero()
    tiles = sorted(set(((bad[:, 1] // 16) % 16).tolist()))
    summary.append((int(len(bad)), round(float(d.max()), 6), tiles[:16]))
print(os.environ.get('TAG', ''), 'B', B, 'N', N, 'bounded', bounded, 'per-launch (wrong points, max err, tile%16):', summary)
