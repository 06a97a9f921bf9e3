"""Floating-point inputs other than fp32 at the tops of the API.

The reference's layer is plain PyTorch and runs in whatever floating dtype its inputs have (fp64 in a gradcheck, fp16 / bf16 under
autocast-style callers).  The kernels compute in fp32 -- accumulations in registers, no other precision exists on the device path -- so
the solver / sampler entry points cast such inputs to fp32 on the way in (differentiably: `.to()` is an autograd node, the gradients
come back in the caller's dtype) and the floating-point outputs back on the way out.  fp32 inputs do not pass through here (one dtype
comparison per call).  Integer / boolean inputs are left alone and fail where they did before.
"""
import copy
import functools

import torch

F32 = torch.float32


def _cast_tensor(v, dtype):
    return v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() and v.dtype != dtype else v


def cast_params(obj, dtype=F32):
    """Shallow copy of a camera / cost-function object whose floating tensors are cast to `dtype` (the original is not touched)."""
    if obj is None:
        return None
    other = obj.shallow_copy() if hasattr(obj, 'shallow_copy') else copy.copy(obj)
    names = tuple(getattr(other, '_batched', ())) or tuple(vars(other))
    for name in names:
        v = getattr(other, name, None)
        if isinstance(v, torch.Tensor):
            setattr(other, name, _cast_tensor(v, dtype))
    return other


def cast_tree(out, dtype):
    if isinstance(out, torch.Tensor):
        return _cast_tensor(out, dtype)
    if isinstance(out, (tuple, list)):
        return type(out)(cast_tree(o, dtype) for o in out)
    return out


def foreign_dtype(*tensors):
    """The caller's floating dtype when it is not fp32 (taken from the first floating tensor), else None."""
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_floating_point():
            return None if t.dtype == F32 else t.dtype
    return None


def fp32_boundary(fn):
    """Method decorator for `fn(self, x3d, x2d, w2d, [pose, ] camera, cost_fun, *args, **kwargs)`-shaped entry points."""
    @functools.wraps(fn)
    def wrapper(self, x3d, x2d, w2d, *args, **kwargs):
        dt = foreign_dtype(x2d, x3d, w2d)
        if dt is None:
            return fn(self, x3d, x2d, w2d, *args, **kwargs)
        conv = lambda v: _cast_tensor(v, F32) if isinstance(v, torch.Tensor) else (
            cast_params(v) if hasattr(v, 'shallow_copy') or hasattr(v, 'cam_mats') or hasattr(v, 'delta') else v)
        out = fn(self, conv(x3d), conv(x2d), conv(w2d), *[conv(a) for a in args], **{k: conv(v) for k, v in kwargs.items()})
        return cast_tree(out, dt)
    return wrapper
