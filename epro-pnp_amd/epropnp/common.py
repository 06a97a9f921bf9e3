"""SE(3) helpers, evaluate_pnp, pnp_normalize / pnp_denormalize.

API mirror of the reference's epropnp/common.py.  PyTorch, autograd-capable; used by the framework-level API and
by the differentiable Gauss-Newton step.  The HIP kernels restate the same math in registers (csrc/pnp_math.h).
"""
import torch


def skew(x):
    """(*,3) -> (*,3,3) cross-product matrices."""
    a, b, c = x.unbind(-1)
    o = torch.zeros_like(a)
    return torch.stack((o, -c, b, c, o, -a, -b, a, o), dim=-1).reshape(x.shape[:-1] + (3, 3))


def quaternion_to_rot_mat(quaternions):
    """(*,4) [w,i,j,k] (assumed unit, not normalised here) -> (*,3,3).
    R = (w^2 - |v|^2) I + 2 v v^T + 2 w [v]x  (reference: epropnp/common.py:21-42)."""
    w = quaternions[..., :1]
    v = quaternions[..., 1:]
    outer = v.unsqueeze(-1) * v.unsqueeze(-2)
    eye = torch.eye(3, dtype=quaternions.dtype, device=quaternions.device)
    scal = (w * w - (v * v).sum(-1, keepdim=True)).unsqueeze(-1)
    return 2 * (w.unsqueeze(-1) * skew(v) + outer) + scal * eye


def yaw_to_rot_mat(yaw):
    """(*) -> (*,3,3) rotation about the Y axis (reference: epropnp/common.py:45-64)."""
    c, s = torch.cos(yaw), torch.sin(yaw)
    o, i = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack((c, o, s, o, i, o, -s, o, c), dim=-1).reshape(yaw.shape + (3, 3))


def pose_rotation(pose):
    return yaw_to_rot_mat(pose[..., 3]) if pose.size(-1) == 4 else quaternion_to_rot_mat(pose[..., 3:])


def evaluate_pnp(x3d, x2d, w2d, pose, camera, cost_fun, out_jacobian=False, out_residual=False, out_cost=False,
                 **kwargs):
    """Project, then apply the robust cost.  Each out_* is False (skip), True (return) or a tensor (filled).
    Returns (residual (*,2n) | None, cost (*) | None, jacobian (*,2n,4|6) | None).
    Reference: epropnp/common.py:67-100."""
    if out_cost is True and out_jacobian is False and out_residual is False and not kwargs and _cost_only_fast_path(
            x3d, x2d, w2d, pose, cost_fun):
        # cost of one pose / a grid of poses per object (AMIS-style broadcast, Det's orientation grid): the cost-only
        # sweep kernel, no (P,B,N,.) temporaries
        from . import functional as hip
        prob = hip.PnPProblem(x3d, x2d, w2d, camera, cost_fun, 4 if pose.size(-1) == 4 else 6)
        return None, hip.evaluate_cost(prob, pose), None
    jac_buf = out_jacobian
    if isinstance(out_jacobian, torch.Tensor):
        jac_buf = out_jacobian.view(x2d.shape[:-1] + (2, out_jacobian.size(-1)))
    x2d_proj, jac_cam = camera.project(x3d, pose, out_jac=jac_buf, **kwargs)
    return cost_fun.compute(x2d_proj, x2d, w2d, jac_cam=jac_cam, out_residual=out_residual, out_cost=out_cost,
                            out_jacobian=out_jacobian)


def _cost_only_fast_path(x3d, x2d, w2d, pose, cost_fun):
    """HIP tensors, nothing to differentiate, points (B,N,.) and poses (B,p) or (P,B,p)."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (x3d, x2d, w2d, pose)):
        return False
    delta = getattr(cost_fun, 'delta', None)
    if isinstance(delta, torch.Tensor) and (delta.requires_grad and torch.is_grad_enabled()):
        return False
    if x3d.dim() != 3 or x2d.dim() != 3 or w2d.dim() != 3 or x3d.size(0) == 0:
        return False
    B = x3d.size(0)
    if not ((pose.dim() == 2 and pose.size(0) == B) or (pose.dim() == 3 and pose.size(1) == B)):
        return False
    from . import _hip
    return _hip.on_hip_path(x3d, x2d, w2d, pose)


def rotate_offset(pose, offset):
    """R(pose) @ offset for broadcastable leading dims, written out element-wise: a batched 3x3 @ 3x1 `matmul` over
    (S, B) poses dispatches a BLAS GEMM per call (measured ~1 ms for 128 x 600 poses on MI355X)."""
    ox, oy, oz = offset.unbind(-1)
    if pose.size(-1) == 4:
        c, s = torch.cos(pose[..., 3]), torch.sin(pose[..., 3])
        return torch.stack((c * ox + s * oz, oy.expand_as(c), c * oz - s * ox), dim=-1)
    rot = quaternion_to_rot_mat(pose[..., 3:])
    return (rot * offset.unsqueeze(-2)).sum(dim=-1)


def pnp_normalize(x3d, pose=None, detach_transformation=True):
    """Centre x3d on its mean; shift the pose translation accordingly.  -> offset (*,3), x3d_norm, pose_norm|None."""
    if detach_transformation and x3d.dim() == 3 and x3d.size(0) > 0 and _fused(x3d):
        from . import functional as hip      # two launches (csrc/eval_kernels.hip) instead of ~12
        offset, x3d_norm = hip.center_points(x3d)
    else:
        offset = (x3d.detach() if detach_transformation else x3d).mean(dim=-2)
        x3d_norm = x3d - offset.unsqueeze(-2)
    if pose is None:
        return offset, x3d_norm, None
    if _can_shift(pose, offset):
        from . import functional as hip
        return offset, x3d_norm, hip.shift_poses(pose, offset, +1.0)
    return offset, x3d_norm, torch.cat((pose[..., :3] + rotate_offset(pose, offset), pose[..., 3:]), dim=-1)


def _fused(*tensors):
    from . import _hip
    return _hip.on_hip_path(*tensors)


def _can_shift(pose, offset):
    """The fused pose shift (differentiable w.r.t. the pose, offset constant): tensors of shape (...,B,pose_len) with
    offset (B,3)."""
    return (not offset.requires_grad and offset.dim() == 2 and pose.dim() >= 2
            and pose.size(-2) == offset.size(0) and pose.numel() > 0 and _fused(pose, offset))


def pnp_denormalize(offset, pose_norm):
    if _can_shift(pose_norm, offset):
        from . import functional as hip
        return hip.shift_poses(pose_norm, offset, -1.0)
    return torch.cat((pose_norm[..., :3] - rotate_offset(pose_norm, offset), pose_norm[..., 3:]), dim=-1)
