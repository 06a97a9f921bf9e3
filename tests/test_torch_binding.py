"""The C++ autograd nodes (csrc/torch_binding.cpp -> lib/_epropnp_torch.so) against the ctypes nodes of
epropnp/functional.py: both call the same entry points of libepropnp_hip.so, so values and gradients are bit-identical."""
import os

import pytest
import torch

import epropnp_oracle as orc
from helpers import make_layer_objects, pack_noise


def test_module_is_built_and_matches_the_abi():
    """No GPU needed: the module loads next to the HIP library and agrees on the ABI version and struct layout."""
    import ctypes
    from epropnp import _hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, 'epro-pnp_amd', 'lib', '_epropnp_torch.so')):
        pytest.skip('torch binding not built (python epro-pnp_amd/build.py)')
    import install as emu
    emu.uninstall()
    _hip._torch_ext = False
    ext = _hip.torch_ext()
    assert ext is not None and ext.abi_version() == _hip.ABI_VERSION and ext.mc_params_size() == ctypes.sizeof(_hip.McParams)


@pytest.fixture
def dev():
    import install as emu
    from epropnp import _hip
    assert torch.cuda.is_available()
    emu.uninstall()
    _hip._torch_ext = False
    yield torch.device('cuda:0')
    _hip._torch_ext = False


def _step(dev, dof, normalize, rslm, plus, use_ext):
    from epropnp import _hip
    from epropnp.epropnp import EProPnP4DoF, EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver, RSLMSolver
    from epropnp.losses import monte_carlo_pose_loss
    _hip._torch_ext = False if use_ext else None          # False: detect again; None: ctypes nodes
    assert (_hip.torch_ext() is not None) == use_ext
    B, N, S, K = 9, 70, 64, 4
    prob = orc.make_problem(B, N, dof, seed=3, bounds='tensor' if dof == 4 else None)
    noise = pack_noise(orc.make_noise(B, S, K, dof, seed=4), dof).to(dev)
    rn = orc.make_rslm_noise(prob, dof, 8, 12, seed=5)
    p, cam, cf = make_layer_objects(prob, dev, relative_delta=0.5)
    x3d, x2d, w2d = (p[k].clone().requires_grad_(True) for k in ('x3d', 'x2d', 'w2d'))
    cf.set_param(x2d, w2d)                                    # x2d NOT detached: the delta node's x2d gradient as well
    init = RSLMSolver(dof=dof, num_points=8, num_proposals=12, num_iter=3) if rslm else None
    if rslm:
        init.draw = lambda w: (rn['inds'].to(dev), rn['rot'].float().to(dev))
    layer = (EProPnP6DoF if dof == 6 else EProPnP4DoF)(mc_samples=S, num_iter=K, normalize=normalize,
                                                       solver=LMSolver(dof=dof, num_iter=3, init_solver=init))
    out = layer.monte_carlo_forward(x3d, x2d, w2d, cam, cf, pose_init=p['pose_init'], force_init_solve=rslm,
                                    with_pose_opt_plus=plus, with_cost=True, noise=noise)
    pose_opt, cost, pplus, samples, logw, cost_init = out
    loss = monte_carlo_pose_loss(logw, cost_init).mean()
    if plus:
        loss = loss + 0.1 * pplus.sum()
    loss.backward()
    return [t.detach().cpu() for t in (pose_opt, cost, samples, logw, cost_init, cf.delta, x3d.grad, x2d.grad, w2d.grad)]


@pytest.mark.gpu
@pytest.mark.parametrize('dof,normalize,rslm,plus', [(6, False, False, False), (4, True, True, True), (6, True, False, True),
                                                     (4, False, True, False)])
def test_cpp_nodes_equal_ctypes_nodes(dev, dof, normalize, rslm, plus):
    a = _step(dev, dof, normalize, rslm, plus, use_ext=True)
    b = _step(dev, dof, normalize, rslm, plus, use_ext=False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_cpp_nodes_guard_and_status(dev):
    from epropnp import _hip
    from epropnp import functional as F
    from epropnp.epropnp import EProPnP6DoF
    from epropnp.levenberg_marquardt import LMSolver
    assert _hip.torch_ext() is not None
    prob = orc.make_problem(4, 40, 6, seed=12)
    p, cam, cf = make_layer_objects(prob, dev, relative_delta=0.5)
    x3d = p['x3d'].clone().requires_grad_(True)
    x2d = p['x2d'].clone()
    cf.set_param(x2d, p['w2d'])
    layer = EProPnP6DoF(mc_samples=32, num_iter=2, solver=LMSolver(dof=6, num_iter=3))
    out = layer.monte_carlo_forward(x3d, x2d, p['w2d'], cam, cf, pose_init=p['pose_init'], force_init_solve=False)
    x2d.add_(1.0)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        (out[5] + torch.logsumexp(out[4], 0)).sum().backward()
    bad = p['x3d'].clone()
    bad[2, 5, 1] = float('nan')
    cf.set_param(p['x2d'], p['w2d'])
    with pytest.raises(RuntimeError, match='object 2'):
        with F.numerics_check():
            layer.monte_carlo_forward(bad, p['x2d'], p['w2d'], cam, cf, pose_init=p['pose_init'], force_init_solve=False)


@pytest.mark.gpu
def test_mc_loss_node_without_cost_target(dev):
    """functional.mc_pose_loss(logweights, None): the C++ node has a single tensor input then; its backward must not ask
    needs_input_grad for an edge that does not exist (it raised 'Index out of range'), and agrees with the ctypes node."""
    from epropnp import _hip
    from epropnp import functional as F
    g = torch.Generator().manual_seed(3)
    logw0 = torch.randn(40, 6, generator=g).to(dev)
    grads = []
    for use_ext in (True, False):
        if not use_ext:
            os.environ['EPROPNP_NO_TORCH_EXT'] = '1'
        _hip._torch_ext = False
        try:
            assert (_hip.torch_ext() is not None) == use_ext
            logw = logw0.clone().requires_grad_(True)
            loss = F.mc_pose_loss(logw, None)
            assert torch.allclose(loss, torch.logsumexp(logw0, 0), atol=1e-5)
            loss.sum().backward()
            grads.append(logw.grad.clone())
        finally:
            os.environ.pop('EPROPNP_NO_TORCH_EXT', None)
    assert torch.equal(grads[0], grads[1])
    assert torch.allclose(grads[0], torch.softmax(logw0, 0), atol=1e-6)
