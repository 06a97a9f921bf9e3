"""Shared batch-shape plumbing for the camera / cost parameter objects.

The reference gives PerspectiveCamera and HuberPnPCost four in-place helpers each
(reshape_/expand_/repeat_/shallow_copy; epropnp/camera.py:167-197, epropnp/cost_fun.py:91-112) that RSLMSolver uses
to replicate parameters over proposals.  Here they are generated from a declaration of which attributes are batched
tensors and how many trailing (non-batch) dims each has.
"""
import torch


class BatchedParams:
    _batched = {}      # attribute name -> number of trailing event dims
    _plain = ()        # non-tensor attributes copied by shallow_copy

    def _map(self, fn):
        for name, ev in self._batched.items():
            v = getattr(self, name, None)
            if isinstance(v, torch.Tensor):
                setattr(self, name, fn(v, ev))
        return self

    def reshape_(self, *batch_shape):
        return self._map(lambda v, ev: v.reshape(*batch_shape, *v.shape[v.dim() - ev:]))

    def expand_(self, *batch_shape):
        return self._map(lambda v, ev: v.expand(*batch_shape, *([-1] * ev)))

    def repeat_(self, *batch_repeat):
        return self._map(lambda v, ev: v.repeat(*batch_repeat, *([1] * ev)))

    def shallow_copy(self):
        other = object.__new__(type(self))
        for name in tuple(self._batched) + tuple(self._plain):
            setattr(other, name, getattr(self, name, None))
        return other
