"""Pinhole camera: projection, bounds clamp and the 2xd Jacobian w.r.t. a local pose perturbation.

API mirror of the reference's epropnp/camera.py (PerspectiveCamera :33-197, project_a :10-18, project_b :21-30).
This module is the framework-level (PyTorch, autograd-capable) statement of the camera model; it is what
`evaluate_pnp` and the differentiable Gauss-Newton step use.  The solver / sampler hot path does not call it:
there the same math runs inside the HIP kernels (csrc/pnp_sweep.h, csrc/pnp_math.h).
"""
import torch

from ._batched import BatchedParams
from .common import pose_rotation


def _homogeneous_pixels(x3d, pose, cam_mats, keep_rotated):
    rot = pose_rotation(pose)
    if keep_rotated:                                  # association of project_a: (R x + t) K^T
        x_rot = torch.matmul(x3d, rot.transpose(-1, -2))
        pix = torch.matmul(x_rot + pose[..., None, :3], cam_mats.transpose(-1, -2))
        return pix, x_rot
    kr = torch.matmul(cam_mats, rot)                  # association of project_b: x (K R)^T + K t
    kt = torch.matmul(cam_mats, pose[..., :3, None]).squeeze(-1)
    return torch.matmul(x3d, kr.transpose(-1, -2)) + kt.unsqueeze(-2), None


def project_a(x3d, pose, cam_mats, z_min: float):
    pix, x_rot = _homogeneous_pixels(x3d, pose, cam_mats, True)
    z = pix[..., 2:3].clamp(min=z_min)
    return pix[..., :2] / z, x_rot, z


def project_b(x3d, pose, cam_mats, z_min: float):
    pix, _ = _homogeneous_pixels(x3d, pose, cam_mats, False)
    z = pix[..., 2:3].clamp(min=z_min)
    return pix[..., :2] / z, z


class PerspectiveCamera(BatchedParams):
    """cam_mats (*,3,3); optional projection bounds lb/ub (float or (*,2)), or img_shape (*,2) in [h, w]."""

    _batched = {'cam_mats': 2, 'lb': 1, 'ub': 1}
    _plain = ('z_min', 'allowed_border')

    def __init__(self, cam_mats=None, z_min=0.1, img_shape=None, allowed_border=200, lb=None, ub=None):
        self.z_min = z_min
        self.allowed_border = allowed_border
        self.set_param(cam_mats, img_shape, lb, ub)

    def set_param(self, cam_mats, img_shape=None, lb=None, ub=None):
        self.cam_mats = cam_mats
        if img_shape is None:
            self.lb, self.ub = lb, ub
        else:   # image rectangle grown by `allowed_border` pixels, in [x, y] order
            self.lb = -0.5 - self.allowed_border
            self.ub = img_shape[..., [1, 0]] + (self.allowed_border - 0.5)

    def _bounds(self):
        if self.lb is None or self.ub is None:
            return None, None
        lb = self.lb.unsqueeze(-2) if isinstance(self.lb, torch.Tensor) else self.lb
        ub = self.ub.unsqueeze(-2) if isinstance(self.ub, torch.Tensor) else self.ub
        return lb, ub

    def project(self, x3d, pose, out_jac=False, clip_jac=True):
        """x3d (*,n,3), pose (*,4|7) -> x2d_proj (*,n,2), jac (*,n,2,4|6) or None.
        `out_jac` may be a preallocated tensor that receives the Jacobian (no-grad use only)."""
        want_jac = out_jac is not False
        if want_jac:
            x2d_proj, x_rot, z = project_a(x3d, pose, self.cam_mats, self.z_min)
        else:
            x2d_proj, z = project_b(x3d, pose, self.cam_mats, self.z_min)
        lb, ub = self._bounds()
        if lb is not None:
            lo = lb if isinstance(lb, torch.Tensor) else x2d_proj.new_tensor(lb)
            hi = ub if isinstance(ub, torch.Tensor) else x2d_proj.new_tensor(ub)
            x2d_proj = torch.minimum(torch.maximum(x2d_proj, lo), hi)
        if not want_jac:
            return x2d_proj, None
        dof = 4 if pose.size(-1) == 4 else 6
        jac = self.project_jacobian(x_rot, z, x2d_proj, out_jac if isinstance(out_jac, torch.Tensor) else None, dof)
        if clip_jac:
            dead = (z == self.z_min).expand_as(x2d_proj)
            if lb is not None:
                dead = dead | (x2d_proj == lb) | (x2d_proj == ub)
            if jac.requires_grad:
                jac = jac.masked_fill(dead.unsqueeze(-1), 0)
            else:
                jac.masked_fill_(dead.unsqueeze(-1), 0)
        return x2d_proj, jac

    def project_jacobian(self, x3d_rot, zcam, x2d_proj, out_jac, dof):
        if dof not in (4, 6):
            raise ValueError('dof must be 4 or 6')
        k = self.cam_mats.unsqueeze(-3)                                   # (*,1,3,3)
        inv_z = zcam.unsqueeze(-1)                                         # (*,n,1,1)
        d_cam = torch.cat((k[..., :2, :2] / inv_z, (k[..., :2, 2:3] - x2d_proj.unsqueeze(-1)) / inv_z), dim=-1)
        if dof == 4:      # yaw about Y moves (x, z) of the rotated point
            lever = torch.stack((x3d_rot[..., 2], -x3d_rot[..., 0]), dim=-1).unsqueeze(-1)
            d_rot = torch.matmul(d_cam[..., ::2], lever)
        else:             # left perturbation: d(R x)/d(rot) = skew(2 R x)  (tangent scale of T(q))
            ax, ay, az = (2 * x3d_rot).unbind(-1)
            zero = torch.zeros_like(ax)
            lever = torch.stack((zero, -az, ay, az, zero, -ax, -ay, ax, zero), dim=-1).reshape(
                x3d_rot.shape[:-1] + (3, 3))
            d_rot = torch.matmul(d_cam, lever)
        jac = torch.cat((d_cam, d_rot), dim=-1)
        if out_jac is not None:
            assert not jac.requires_grad, 'out_jac is not supported for backward'
            out_jac.copy_(jac)
            return out_jac
        return jac

    @staticmethod
    def get_quaternion_transfrom_mat(quaternions):
        """(*,4) unit quaternion [w,i,j,k] -> (*,4,3) map from a 3-D tangent step to a quaternion increment."""
        w, i, j, k = quaternions.unbind(-1)
        return torch.stack((i, j, k, -w, -k, j, k, -w, -i, -j, i, -w), dim=-1).reshape(
            quaternions.shape[:-1] + (4, 3))
