"""
oracle/mmdet_shim.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (build container only).

The reference's detection loss (EPro-PnP-Det/epropnp_det/models/losses/monte_carlo_pose_loss.py:8-9) imports
`mmdet.core.reduce_mean`, `mmdet.models.LOSSES` and `mmdet.models.weighted_loss` (pinned mmdet==2.19.1,
EPro-PnP-Det/requirements.txt:7).  mmdet is not installed here and there is no network, so `install()` registers
stand-in modules providing exactly those three names, restating mmdet 2.19.1's published behaviour:

  mmdet/models/losses/utils.py   reduce_loss: 'none' -> loss, 'mean' -> loss.mean(), 'sum' -> loss.sum()
                                 weight_reduce_loss(loss, weight, reduction, avg_factor):
                                     loss *= weight (if given); avg_factor None -> reduce_loss;
                                     else 'mean' -> loss.sum() / avg_factor, 'none' -> loss, 'sum' -> ValueError
                                 weighted_loss(f)(pred, target, weight=None, reduction='mean', avg_factor=None, **kw)
                                     = weight_reduce_loss(f(pred, target, **kw), weight, reduction, avg_factor)
  mmdet/core/utils/dist_utils.py reduce_mean(t): t if torch.distributed is not initialised, else all-reduce mean
  mmdet/models/builder.py        LOSSES: an mmcv Registry; only `.register_module()` (a decorator) is used

"parity unpinned" at this third-party boundary (as for pyro, oracle/pyro_shim.py): the formulas above are mmdet's
documented ones; everything on the reference's side of the boundary runs unmodified.
"""
import functools
import sys
import types

import torch


def reduce_loss(loss, reduction):
    if reduction == 'none':
        return loss
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise ValueError(reduction)


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def weighted_loss(loss_func):
    @functools.wraps(loss_func)
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None, **kwargs):
        return weight_reduce_loss(loss_func(pred, target, **kwargs), weight, reduction, avg_factor)
    return wrapper


def reduce_mean(tensor):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


class _Registry:
    def __init__(self, name):
        self.name, self.module_dict = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco if module is None else deco(module)


def install():
    if 'mmdet' in sys.modules:
        return
    mmdet, core, models = types.ModuleType('mmdet'), types.ModuleType('mmdet.core'), types.ModuleType('mmdet.models')
    core.reduce_mean = reduce_mean
    models.LOSSES, models.weighted_loss = _Registry('loss'), weighted_loss
    mmdet.core, mmdet.models = core, models
    sys.modules.update({'mmdet': mmdet, 'mmdet.core': core, 'mmdet.models': models})
