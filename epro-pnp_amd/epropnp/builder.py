"""Dict-config construction of the layer, for EPro-PnP-Det style configs.

Mirror of EPro-PnP-Det/epropnp_det/ops/pnp/builder.py:7-19 (`build_pnp / build_camera / build_cost_fun` over mmcv
registries).  mmcv is not a dependency here: a minimal registry with the same `dict(type=..., **kwargs)` convention
is used, and nested `solver=` / `init_solver=` dicts are built recursively
(EPro-PnP-Det/epropnp_det/ops/pnp/epropnp.py:54-70).
"""
from .camera import PerspectiveCamera
from .cost_fun import AdaptiveHuberPnPCost, HuberPnPCost
from .epropnp import EProPnP4DoF, EProPnP6DoF
from .levenberg_marquardt import LMSolver, RSLMSolver

PNP = {c.__name__: c for c in (EProPnP4DoF, EProPnP6DoF, LMSolver, RSLMSolver)}
CAMERA = {'PerspectiveCamera': PerspectiveCamera}
COSTFUN = {c.__name__: c for c in (HuberPnPCost, AdaptiveHuberPnPCost)}


def _build(cfg, registry, default_args):
    if cfg is None or not isinstance(cfg, dict):
        return cfg
    args = dict(default_args or {})
    args.update(cfg)
    kind = args.pop('type')
    cls = registry[kind] if isinstance(kind, str) else kind
    if isinstance(args.get('init_solver'), dict):      # nested RSLMSolver inherits the parent's dof
        sub = dict(args['init_solver'])
        sub.setdefault('dof', args.get('dof', 4))
        args['init_solver'] = _build(sub, PNP, None)
    # a nested `solver=dict(...)` is built by EProPnPBase.__init__ itself (it knows its dof)
    return cls(**args)


def build_pnp(cfg, **default_args):
    return _build(cfg, PNP, default_args)


def build_camera(cfg, **default_args):
    return _build(cfg, CAMERA, default_args)


def build_cost_fun(cfg, **default_args):
    return _build(cfg, COSTFUN, default_args)
