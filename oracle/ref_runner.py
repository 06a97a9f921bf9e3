"""
oracle/ref_runner.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (build container only).

Runs the UNMODIFIED reference (`/root/reference/epropnp`) on CPU with all of its random draws replaced by
injected tensors, so that the reference, the restatement (oracle/epropnp_oracle.py) and the HIP kernels can be
compared on identical noise.  Nothing here is importable on the GPU box (no /root/reference there); the
fixtures it produces are committed under tests/golden/ by oracle/make_golden.py.

Injection points (reference file:line):
  * pyro MultivariateStudentT.rsample                      <- epropnp/epropnp.py:146
  * epropnp.distributions._standard_normal (ACG.rsample)   <- epropnp/distributions.py:42-46
  * VonMisesUniformMix.sample                              <- epropnp/distributions.py:61-72
  * torch.multinomial / torch.rand / torch.randn in RSLM   <- epropnp/levenberg_marquardt.py:306-323
"""
import importlib
import os
import sys
import types

import torch

REF_ROOT = os.environ.get('EPROPNP_REFERENCE', '/root/reference')
_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import pyro_shim  # noqa: E402
import epropnp_oracle as orc  # noqa: E402


class _Inject:
    def __init__(self):
        self.reset(None, None)

    def reset(self, noise, rslm):
        self.noise, self.rslm = noise, rslm
        self.it_t = self.it_r = 0


INJ = _Inject()
_ref = {}


def load_reference():
    """Import the reference package (once) and patch its random draws."""
    if _ref:
        return _ref
    assert os.path.isdir(os.path.join(REF_ROOT, 'epropnp')), 'reference checkout not found'
    pyro_shim.install()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    assert 'epropnp' not in sys.modules, 'a different `epropnp` package is already imported'
    mods = {n: importlib.import_module('epropnp.' + n)
            for n in ('common', 'camera', 'cost_fun', 'distributions', 'levenberg_marquardt', 'epropnp')}
    assert mods['epropnp'].__file__.startswith(REF_ROOT)
    from pyro.distributions import MultivariateStudentT

    def t_rsample(self, sample_shape=torch.Size()):
        z, chi2 = INJ.noise['z'][INJ.it_t], INJ.noise['chi2'][INJ.it_t]
        INJ.it_t += 1
        y = z * torch.rsqrt(chi2 / self.df).unsqueeze(-1)
        return self.loc + self.scale_tril.matmul(y.unsqueeze(-1)).squeeze(-1)
    MultivariateStudentT.rsample = t_rsample

    def std_normal(shape, dtype, device):
        g = INJ.noise['g'][INJ.it_r]
        INJ.it_r += 1
        assert tuple(g.shape) == tuple(shape), (g.shape, shape)
        return g
    mods['distributions']._standard_normal = std_normal

    _ref['vm_sample_unpatched'] = mods['distributions'].VonMisesUniformMix.sample      # numpy-driven original (:61-72)

    def vm_sample(self, sample_shape=torch.Size()):
        u, vm = INJ.noise['u'][INJ.it_r], INJ.noise['vm'][INJ.it_r]
        INJ.it_r += 1
        return orc.vm_mix_sample(self.loc, self.concentration, u, vm, sample_shape[0], self.uniform_mix)
    mods['distributions'].VonMisesUniformMix.sample = vm_sample

    class _TorchProxy(types.ModuleType):
        def __getattr__(self, name):
            return getattr(torch, name)

    proxy = _TorchProxy('torch_proxy')

    def multinomial(weights, num_samples, **kw):          # (P*B, N) -> (P*B, n)
        return INJ.rslm['inds'].reshape(-1, num_samples).clone()

    def rand(shape, **kw):
        return (INJ.rslm['rot'] / (2 * torch.pi)).reshape(shape).clone()   # reference multiplies by 2 pi

    def randn(shape, **kw):
        return INJ.rslm['rot'].reshape(shape).clone()                      # already unit; re-normalised :324-325
    proxy.multinomial, proxy.rand, proxy.randn = multinomial, rand, randn
    mods['levenberg_marquardt'].torch = proxy
    _ref.update(mods)
    return _ref


def build_layer(dof, mc_samples, num_iter, lm_iter, normalize=False, rslm=None):
    m = load_reference()
    LM, RS = m['levenberg_marquardt'].LMSolver, m['levenberg_marquardt'].RSLMSolver
    init = RS(dof=dof, **rslm) if rslm else None
    solver = LM(dof=dof, num_iter=lm_iter, init_solver=init)
    cls = m['epropnp'].EProPnP6DoF if dof == 6 else m['epropnp'].EProPnP4DoF
    return cls(mc_samples=mc_samples, num_iter=num_iter, normalize=normalize, solver=solver)


def make_camera(prob):
    m = load_reference()
    return m['camera'].PerspectiveCamera(cam_mats=prob['cam_mats'], z_min=0.1, lb=prob.get('lb'), ub=prob.get('ub'))


def run_lm(prob, dof, lm_iter, fast_mode=False):
    """LMSolver.solve with pose_init -> pose_opt, pose_cov, cost."""
    m = load_reference()
    solver = m['levenberg_marquardt'].LMSolver(dof=dof, num_iter=lm_iter)
    cf = m['cost_fun'].HuberPnPCost(delta=prob['delta'])
    return solver.solve(prob['x3d'], prob['x2d'], prob['w2d'], make_camera(prob), cf, pose_init=prob['pose_init'],
                        with_pose_cov=True, with_cost=True, fast_mode=fast_mode)


def run_evaluate(prob, pose, jac=False, clip_jac=True):
    m = load_reference()
    cf = m['cost_fun'].HuberPnPCost(delta=prob['delta'])
    kw = dict(clip_jac=clip_jac) if jac else {}
    with torch.no_grad():
        return m['common'].evaluate_pnp(prob['x3d'], prob['x2d'], prob['w2d'], pose, make_camera(prob), cf,
                                        out_jacobian=jac, out_residual=jac, out_cost=True, **kw)


def run_mc(prob, noise, dof, mc_samples, num_iter, lm_iter, normalize=False, relative_delta=0.5,
           rslm=None, rslm_noise=None, with_pose_opt_plus=False, fast_mode=False, detach_x2d_for_delta=True):
    """monte_carlo_forward + MC loss + backward on the reference.  Returns dict of outputs and input grads."""
    m = load_reference()
    INJ.reset(noise, rslm_noise)
    layer = build_layer(dof, mc_samples, num_iter, lm_iter, normalize, rslm)
    x3d, x2d, w2d = (prob[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cam = make_camera(prob)
    cf = m['cost_fun'].AdaptiveHuberPnPCost(relative_delta=relative_delta)
    cf.set_param(x2d.detach() if detach_x2d_for_delta else x2d, w2d)
    out = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=prob['pose_init'],
                                    force_init_solve=rslm is not None, with_pose_opt_plus=with_pose_opt_plus,
                                    with_cost=True, fast_mode=fast_mode)
    pose_opt, cost, pose_opt_plus, pose_samples, logw, cost_init = out
    loss_obj = cost_init + torch.logsumexp(logw, dim=0)
    loss = loss_obj.mean()
    total = loss
    if with_pose_opt_plus:   # a fixed linear functional so that its gradient is testable
        total = total + 0.1 * (pose_opt_plus * torch.linspace(0.5, 1.5, pose_opt_plus.shape[-1])).sum(-1).mean()
    total.backward()
    res = dict(pose_opt=pose_opt, cost=cost, pose_samples=pose_samples, logweights=logw, cost_init=cost_init,
               loss_obj=loss_obj, delta=cf.delta, gx3d=x3d.grad, gx2d=x2d.grad, gw2d=w2d.grad)
    if with_pose_opt_plus:
        res['pose_opt_plus'] = pose_opt_plus
    return {k: v.detach().clone() for k, v in res.items()}
