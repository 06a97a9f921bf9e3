"""
oracle/pyro_shim.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference imports `pyro` (pinned pyro-ppl==1.6.0, EPro-PnP-Det/requirements.txt:6; call sites
epropnp/epropnp.py:10, epropnp/distributions.py:11-12).  pyro is not installed in this image and there is
no network, so `install()` registers a minimal stand-in that provides exactly the four names the reference
uses.  `MultivariateStudentT` restates pyro 1.6.0's published arithmetic:

    rsample : loc + scale_tril @ (N(0,I) * rsqrt(Chi2(df) / df))
    log_prob: -0.5 (df+n) log1p(|L^-1 (x-loc)|^2 / df)
              - [sum log diag L + 0.5 n log df + 0.5 n log pi + lgamma(df/2) - lgamma((df+n)/2)]

No reference test pins these values ("parity unpinned" at this third-party boundary);
the formulas are the standard multivariate Student-t.

Used only by oracle/ref_runner.py / oracle/make_golden.py in the build container.
"""
import math
import sys
import types

import torch
from torch.distributions import Chi2, constraints
from torch.distributions.distribution import Distribution
from torch.distributions.multivariate_normal import _batch_mahalanobis
from torch.distributions.utils import lazy_property


class MultivariateStudentT(Distribution):
    arg_constraints = {'df': constraints.positive, 'loc': constraints.real_vector,
                       'scale_tril': constraints.lower_cholesky}
    support = constraints.real_vector
    has_rsample = True

    def __init__(self, df, loc, scale_tril, validate_args=None):
        dim = loc.size(-1)
        assert scale_tril.shape[-2:] == (dim, dim)
        if not isinstance(df, torch.Tensor):
            df = loc.new_tensor(df)
        batch_shape = torch.broadcast_shapes(df.shape, loc.shape[:-1], scale_tril.shape[:-2])
        event_shape = torch.Size((dim,))
        self.df = df.expand(batch_shape)
        self.loc = loc.expand(batch_shape + event_shape)
        self._unbroadcasted_scale_tril = scale_tril
        self._chi2 = Chi2(self.df)
        super().__init__(batch_shape, event_shape, validate_args=validate_args)

    @lazy_property
    def scale_tril(self):
        return self._unbroadcasted_scale_tril.expand(self._batch_shape + self._event_shape + self._event_shape)

    def rsample(self, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        X = torch.empty(shape, dtype=self.df.dtype, device=self.df.device).normal_()
        Z = self._chi2.rsample(sample_shape)
        Y = X * torch.rsqrt(Z / self.df).unsqueeze(-1)
        return self.loc + self.scale_tril.matmul(Y.unsqueeze(-1)).squeeze(-1)

    def log_prob(self, value):
        n = self.loc.size(-1)
        y = _batch_mahalanobis(self._unbroadcasted_scale_tril, value - self.loc)
        Z = (self._unbroadcasted_scale_tril.diagonal(dim1=-2, dim2=-1).log().sum(-1)
             + 0.5 * n * self.df.log() + 0.5 * n * math.log(math.pi)
             + torch.lgamma(0.5 * self.df) - torch.lgamma(0.5 * (self.df + n)))
        return -0.5 * (self.df + n) * torch.log1p(y / self.df) - Z


def install():
    """Register the stand-in as `pyro`, `pyro.distributions`, `pyro.distributions.util` unless real pyro exists."""
    try:
        import pyro  # noqa: F401
        return False
    except ImportError:
        pass
    pyro = types.ModuleType('pyro')
    dist = types.ModuleType('pyro.distributions')
    util = types.ModuleType('pyro.distributions.util')
    dist.MultivariateStudentT = MultivariateStudentT
    dist.TorchDistribution = Distribution
    dist.constraints = constraints
    util.broadcast_shape = torch.broadcast_shapes
    dist.util = util
    pyro.distributions = dist
    sys.modules['pyro'] = pyro
    sys.modules['pyro.distributions'] = dist
    sys.modules['pyro.distributions.util'] = util
    return True
