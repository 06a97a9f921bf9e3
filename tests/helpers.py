"""Shared test helpers: golden fixtures, oracle <-> kernel noise layout, problem construction."""
import os

import numpy as np
import torch

import epropnp_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    out = {}
    for k in z.files:
        v = torch.from_numpy(z[k]) if z[k].ndim > 0 else z[k].item()
        if '.' in k:
            a, b = k.split('.', 1)
            out.setdefault(a, {})[b] = v
        else:
            out[k] = v
    return out


def pack_noise(noise, dof):
    """oracle noise dict ((K,s,B,.) tensors) -> kernel layout (B,K,s,stride)."""
    z, chi2 = noise['z'], noise['chi2']
    K, s, B, _ = z.shape
    if dof == 6:
        flat = torch.cat((z, chi2.unsqueeze(-1), noise['g']), -1)                    # (K,s,B,8)
    else:
        T = orc.VM_MAX_TRIES
        tail = torch.zeros(K, s, B, 3 * T, dtype=z.dtype)
        n_u = noise['u'].shape[1]
        tail[:, :n_u, :, 0] = noise['u'][..., 0]
        tail[:, n_u:] = noise['vm'].reshape(K, s - n_u, B, 3 * T)
        flat = torch.cat((z, chi2.unsqueeze(-1), tail), -1)
    return flat.permute(2, 0, 1, 3).contiguous().float()


def to_dev(d, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def make_layer_objects(prob, device, relative_delta=None):
    """reference-API camera / cost_fun objects for a problem dict (from a fixture or orc.make_problem)."""
    from epropnp.camera import PerspectiveCamera
    from epropnp.cost_fun import AdaptiveHuberPnPCost, HuberPnPCost
    p = to_dev(prob, device)
    cam = PerspectiveCamera(cam_mats=p['cam_mats'], z_min=float(p.get('z_min', 0.1)), lb=p.get('lb'), ub=p.get('ub'))
    if relative_delta is None:
        cf = HuberPnPCost(delta=p['delta'])
    else:
        cf = AdaptiveHuberPnPCost(relative_delta=relative_delta)
    return p, cam, cf


def assert_within_spread(err, spread, bar, k=2.0, what=''):
    """Parity bar with the reference's own rounding sensitivity as the yardstick, compared as ORDER STATISTICS over the
    objects of the batch: the i-th largest error must not exceed `bar + k x` the i-th largest spread.

    `spread` (orc.rounding_spread / the fixtures' `spread.*`) is how far the reference's own fp32 result moves per object
    when its inputs move by <= 3 ulp (plus its fp32-vs-fp64 drift).  It is ~0 for well-conditioned objects and large
    where a trust-region decision sits on a knife edge or the LM valley is flat -- but WHICH object of a batch trips
    depends on the individual roundings, so errors and spreads are matched by rank, not by object index.  HOW MANY trip
    is a count of rare independent events: if `n_trip` objects have a spread above the bar, two correct implementations
    differ in that count by ~sqrt(2 n_trip), so the error of rank i is held against the spread of rank
    i - (1 + ceil(sqrt(2 n_trip))); with no such object (n_trip = 0) the match is strict."""
    import math
    e = torch.sort(err.detach().flatten().double().cpu(), descending=True).values
    s = torch.sort(spread.detach().flatten().double().cpu(), descending=True).values
    assert e.numel() == s.numel(), (e.shape, s.shape)
    n_trip = int((s > bar).sum())
    slack = 0 if n_trip == 0 else 1 + math.ceil(math.sqrt(2 * n_trip))
    idx = (torch.arange(e.numel()) - slack).clamp(min=0)
    lim = bar + k * s[idx]
    bad = e > lim
    assert not bool(bad.any()), (f'{what}: {int(bad.sum())} of {e.numel()} order statistics above bar {bar:g} + {k:g} x spread '
                                 f'(rank slack {slack}); worst: err {e[bad][0].item():.3e} vs limit {lim[bad][0].item():.3e} '
                                 f'(rank {int(bad.nonzero()[0])})')

def rel_per_object(a, b):
    """max |a - b| over an object's entries relative to the object's largest |b| -> (B,)"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = (a - b).abs().reshape(a.shape[0], -1).amax(1)
    return d / b.abs().reshape(b.shape[0], -1).amax(1).clamp(min=1e-30)


def set_tune(monkeypatch, **keys):
    """EPROPNP_TUNE (csrc/pnp_host.h: tune_value, epropnp/_hip.py: tune) from keyword arguments: `bwd_impl='valu'`, `bwd_mfma='4,2'`,
    a bare flag as `rslm_composite=True`; no keys (or only None / False values) removes the variable."""
    items = [k if v is True else f'{k}={v}' for k, v in keys.items() if v is not None and v is not False]
    if items:
        monkeypatch.setenv('EPROPNP_TUNE', ';'.join(items))
    else:
        monkeypatch.delenv('EPROPNP_TUNE', raising=False)
