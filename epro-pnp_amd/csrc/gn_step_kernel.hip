// gn_step_kernel.hip -- the differentiable Gauss-Newton step of the derivative-regularisation loss, fused.
//
// Replaces LMSolver.gn_step (epropnp/levenberg_marquardt.py:243-253):
//     residual, _, jac = evaluate_pnp(..., out_jacobian=True, out_residual=True)      # (B,2N), (B,2N,d), clip_jac on
//     step = -solve(jac^T jac + eps I, jac^T residual)
// and what autograd builds for it.  The reference materialises the (B,2N,d) Jacobian with autograd history and runs
// batched bmm / LU kernels forward and backward (58 ms at B=4096, N=512 on MI355X through ATen); here the forward is
// one sweep + an in-register solve, and the backward is two sweeps over register-resident points:
//   with A = J^T J + eps I, b = J^T r, step = -A^-1 b and upstream g = dL/dstep, lambda = A^-1 g:
//     dL = -lambda^T (db + dA step)  =>  per residual row (J_row in R^d, r_row):
//     dL/dJ_row = -(lambda (r_row + J_row.step) + step (J_row.lambda)),   dL/dr_row = -(J_row.lambda)
//   followed by the hand-derived backward of the per-point projection / Jacobian / Huber rescaling
//   (camera.py:10-18,81-143, cost_fun.py:45-84; pose is not differentiated, as in the reference).
#include "dispatch.h"
#include "pnp_host.h"

namespace pnp {

// forward intermediates of one point on the Jacobian path, kept for the hand-written backward
template <int DOF>
struct PointFwd {
  float xr0, xr1, xr2, hx, hy, hz, rz, ppx, ppy, px, py;
  float d0[3], d1[3];        // d x2d / d x3d_cam rows
  float Jc0[DOF], Jc1[DOF];  // camera Jacobian rows (before clip / rescaling)
  float dx, dy, rx, ry, rho, rc, gam, s0, s1, e0, e1;
  bool k0, k1, outlier, rho_live;
};

template <int DOF, bool BOUNDS>
PNP_FN void point_forward(const Point& p, const float (&K)[9], const float (&R)[9], const float* t, float z_min, float delta,
                          float huber_eps, const Bounds& bd, PointFwd<DOF>& f) {
  f.xr0 = R[0] * p.X + R[1] * p.Y + R[2] * p.Z;
  f.xr1 = R[3] * p.X + R[4] * p.Y + R[5] * p.Z;
  f.xr2 = R[6] * p.X + R[7] * p.Y + R[8] * p.Z;
  const float c0 = f.xr0 + t[0], c1 = f.xr1 + t[1], c2 = f.xr2 + t[2];
  f.hx = c0 * K[0] + c1 * K[1] + c2 * K[2];
  f.hy = c0 * K[3] + c1 * K[4] + c2 * K[5];
  f.hz = c0 * K[6] + c1 * K[7] + c2 * K[8];
  const float z = fmaxf(f.hz, z_min);
  f.rz = 1.0f / z;
  f.ppx = f.hx * f.rz;
  f.ppy = f.hy * f.rz;
  f.px = f.ppx;
  f.py = f.ppy;
  if (BOUNDS) {
    f.px = clamp_lu(f.px, bd.lbx, bd.ubx);
    f.py = clamp_lu(f.py, bd.lby, bd.uby);
  }
  f.d0[0] = K[0] * f.rz; f.d0[1] = K[1] * f.rz; f.d0[2] = (K[2] - f.px) * f.rz;
  f.d1[0] = K[3] * f.rz; f.d1[1] = K[4] * f.rz; f.d1[2] = (K[5] - f.py) * f.rz;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f.Jc0[i] = f.d0[i];
    f.Jc1[i] = f.d1[i];
  }
  if (DOF == 6) {
    const float ax = 2.f * f.xr0, ay = 2.f * f.xr1, az = 2.f * f.xr2;
    f.Jc0[3] = f.d0[1] * az - f.d0[2] * ay;
    f.Jc0[4] = f.d0[2] * ax - f.d0[0] * az;
    f.Jc0[5] = f.d0[0] * ay - f.d0[1] * ax;
    f.Jc1[3] = f.d1[1] * az - f.d1[2] * ay;
    f.Jc1[4] = f.d1[2] * ax - f.d1[0] * az;
    f.Jc1[5] = f.d1[0] * ay - f.d1[1] * ax;
  } else {
    f.Jc0[3] = f.d0[0] * f.xr2 - f.d0[2] * f.xr0;
    f.Jc1[3] = f.d1[0] * f.xr2 - f.d1[2] * f.xr0;
  }
  f.dx = f.px - p.u;
  f.dy = f.py - p.v;
  f.rx = f.dx * p.wu;
  f.ry = f.dy * p.wv;
  f.rho = sqrtf(f.rx * f.rx + f.ry * f.ry);
  f.rc = fmaxf(f.rho, huber_eps);                             // cost_fun.py:18-22: max(rho, eps)
  f.rho_live = f.rho >= huber_eps;                            // the clamp passes the gradient only where it is inactive
  f.outlier = f.rc > delta;                                   // min(delta / max(rho, eps), 1) < 1
  f.gam = f.outlier ? sqrtf(delta / f.rc) : 1.0f;
  const bool zc = (z == z_min);
  f.k0 = zc;
  f.k1 = zc;
  if (BOUNDS) {
    f.k0 = f.k0 || (f.px == bd.lbx) || (f.px == bd.ubx);
    f.k1 = f.k1 || (f.py == bd.lby) || (f.py == bd.uby);
  }
  f.s0 = f.k0 ? 0.f : p.wu * f.gam;
  f.s1 = f.k1 ? 0.f : p.wv * f.gam;
  f.e0 = f.rx * f.gam;
  f.e1 = f.ry * f.gam;
}

// accumulate J^T J (upper), J^T r of one point from its forward record
template <int DOF>
PNP_FN void accumulate_normal_eq(const PointFwd<DOF>& f, float (&acc)[NormalEq<DOF>::NV]) {
  float J0[DOF], J1[DOF];
#pragma unroll
  for (int i = 0; i < DOF; ++i) {
    J0[i] = f.s0 * f.Jc0[i];
    J1[i] = f.s1 * f.Jc1[i];
  }
  int idx = 0;
#pragma unroll
  for (int i = 0; i < DOF; ++i)
#pragma unroll
    for (int j = i; j < DOF; ++j) {
      acc[idx] = fmaf(J0[i], J0[j], fmaf(J1[i], J1[j], acc[idx]));
      ++idx;
    }
#pragma unroll
  for (int i = 0; i < DOF; ++i)
    acc[NormalEq<DOF>::NH + i] = fmaf(J0[i], f.e0, fmaf(J1[i], f.e1, acc[NormalEq<DOF>::NH + i]));
}

template <int DOF>
PNP_FN void unpack_sym(const float (&acc)[NormalEq<DOF>::NV], float eps, float (&H)[DOF][DOF]) {
  int idx = 0;
#pragma unroll
  for (int i = 0; i < DOF; ++i)
#pragma unroll
    for (int j = i; j < DOF; ++j) {
      H[i][j] = acc[idx];
      H[j][i] = acc[idx];
      ++idx;
    }
#pragma unroll
  for (int i = 0; i < DOF; ++i) H[i][i] += eps;
}

// ------------------------------------------------------------------------------------------------------------
// Workgroup = one object; lanes stride over its points (the sweeps are L2/HBM-bound and launch-latency-sized, so the
// points are simply re-read for the second backward sweep instead of being pinned in registers).
constexpr int kGnMaxThreads = 256;

template <int DOF, bool BOUNDS>
__global__ __launch_bounds__(kGnMaxThreads) void gn_step_forward_kernel(Problem p, float eps, const float* __restrict__ pose,
                                                                      float* __restrict__ step_out,
                                                                      float* __restrict__ pose_plus_out) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NH = NormalEq<DOF>::NH, NV = NormalEq<DOF>::NV;
  PNP_DYN_SMEM(float, scratch);          // waves * kSumTStride<NV> floats (transposed reduction, <= 4 waves)
  const int b = object_of_block(p.B);
  if (b >= p.B) return;
  float K[9], R[9], ps[PL], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, K, bd, delta);
#pragma unroll
  for (int i = 0; i < PL; ++i) ps[i] = pose[(size_t)b * PL + i];
  pose_to_rot<DOF>(ps, R);
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
#pragma unroll 1
  for (int n0 = 0; n0 < p.N; n0 += (int)blockDim.x) {
    const Point q = load_point(p, b, n0 + (int)threadIdx.x);
    PointFwd<DOF> f;
    point_forward<DOF, BOUNDS>(q, K, R, ps, p.z_min, delta, p.huber_eps, bd, f);
    accumulate_normal_eq<DOF>(f, acc);
  }
  block_sum_t<NV>(acc, scratch);
  float H[DOF][DOF], g[DOF];
  ScaledFactor<DOF> fac;
  unpack_sym<DOF>(acc, eps, H);
#pragma unroll
  for (int i = 0; i < DOF; ++i) g[i] = acc[NH + i];
  const bool spd = scaled_cholesky<DOF>(H, fac);
  scaled_solve<DOF>(fac, g);
  if (threadIdx.x == 0) {
    float st[DOF];
    int st_bits = spd ? 0 : EPROPNP_ST_LM_NOT_SPD;
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
      st[i] = -g[i];
      st_bits |= isfinite(st[i]) ? 0 : EPROPNP_ST_NONFINITE_POSE;
    }
    raise_status(p, st_bits, b);
    if (step_out != nullptr) {
#pragma unroll
      for (int i = 0; i < DOF; ++i) step_out[(size_t)b * DOF + i] = st[i];
    }
    if (pose_plus_out != nullptr) {        // pose_opt_plus = pose (+) step  (LMSolver.forward, :70-72)
      float pp[PL];
      pose_add<DOF>(ps, st, pp);
#pragma unroll
      for (int i = 0; i < PL; ++i) pose_plus_out[(size_t)b * PL + i] = pp[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
template <int DOF, bool BOUNDS>
__global__ __launch_bounds__(kGnMaxThreads) void gn_step_backward_kernel(Problem p, float eps, const float* __restrict__ pose,
                                                                       const float* __restrict__ gstep,
                                                                       const float* __restrict__ gplus,
                                                                       float* __restrict__ gx3d, float* __restrict__ gx2d,
                                                                       float* __restrict__ gw2d, float* __restrict__ gdelta) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NH = NormalEq<DOF>::NH, NV = NormalEq<DOF>::NV;
  PNP_DYN_SMEM(float, scratch);          // waves * kSumTStride<NV> floats
  const int b = object_of_block(p.B);
  if (b >= p.B) return;
  const int T = (int)blockDim.x, tid = (int)threadIdx.x;
  float K[9], R[9], ps[PL], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, K, bd, delta);
#pragma unroll
  for (int i = 0; i < PL; ++i) ps[i] = pose[(size_t)b * PL + i];
  pose_to_rot<DOF>(ps, R);

  // sweep 1: A, b  ->  step = -A^-1 b,  lambda = A^-1 g
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
#pragma unroll 1
  for (int n0 = 0; n0 < p.N; n0 += T) {
    const Point q = load_point(p, b, n0 + tid);
    PointFwd<DOF> f;
    point_forward<DOF, BOUNDS>(q, K, R, ps, p.z_min, delta, p.huber_eps, bd, f);
    accumulate_normal_eq<DOF>(f, acc);
  }
  block_sum_t<NV>(acc, scratch);
  float H[DOF][DOF], step[DOF], lam[DOF];
  ScaledFactor<DOF> fac;
  unpack_sym<DOF>(acc, eps, H);
  scaled_cholesky<DOF>(H, fac);
#pragma unroll
  for (int i = 0; i < DOF; ++i) step[i] = acc[NH + i];
  scaled_solve<DOF>(fac, step);
#pragma unroll
  for (int i = 0; i < DOF; ++i) step[i] = -step[i];
  if (gplus != nullptr) {      // upstream gradient arrives at pose_opt_plus: pull it back through pose_add first
    float gp[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) gp[i] = gplus[(size_t)b * PL + i];
    pose_add_adjoint<DOF>(ps, step, gp, lam);
  } else {
#pragma unroll
    for (int i = 0; i < DOF; ++i) lam[i] = gstep[(size_t)b * DOF + i];
  }
  scaled_solve<DOF>(fac, lam);

  // sweep 2: per-point backward
  float gd = 0.f;
#pragma unroll 1
  for (int n0 = 0; n0 < p.N; n0 += T) {
    const Point q = load_point(p, b, n0 + tid);
    PointFwd<DOF> f;
    point_forward<DOF, BOUNDS>(q, K, R, ps, p.z_min, delta, p.huber_eps, bd, f);
    // row-level gradients
    float a0 = 0.f, c0 = 0.f, a1 = 0.f, c1 = 0.f;
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
      a0 = fmaf(f.s0 * f.Jc0[i], lam[i], a0);
      c0 = fmaf(f.s0 * f.Jc0[i], step[i], c0);
      a1 = fmaf(f.s1 * f.Jc1[i], lam[i], a1);
      c1 = fmaf(f.s1 * f.Jc1[i], step[i], c1);
    }
    float GJ0[DOF], GJ1[DOF];      // dL/d(camera Jacobian rows) after the s0/s1 scaling
    float Gs0 = 0.f, Gs1 = 0.f;
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
      const float gj0 = -(lam[i] * (f.e0 + c0) + step[i] * a0);
      const float gj1 = -(lam[i] * (f.e1 + c1) + step[i] * a1);
      Gs0 = fmaf(gj0, f.Jc0[i], Gs0);
      Gs1 = fmaf(gj1, f.Jc1[i], Gs1);
      GJ0[i] = f.s0 * gj0;
      GJ1[i] = f.s1 * gj1;
    }
    if (f.k0) Gs0 = 0.f;
    if (f.k1) Gs1 = 0.f;
    const float Ge0 = -a0, Ge1 = -a1;
    const float Ggam = f.rx * Ge0 + f.ry * Ge1 + q.wu * Gs0 + q.wv * Gs1;
    float Gwu = f.gam * Gs0, Gwv = f.gam * Gs1;
    float Grx = f.gam * Ge0, Gry = f.gam * Ge1;
    if (f.outlier) {
      gd = fmaf(0.5f * f.gam / delta, Ggam, gd);
      if (f.rho_live) {
        const float Grho = -0.5f * f.gam / f.rho * Ggam;
        Grx = fmaf(Grho, f.rx / f.rho, Grx);
        Gry = fmaf(Grho, f.ry / f.rho, Gry);
      }
    }
    float Gpx = Grx * q.wu, Gpy = Gry * q.wv;
    Gwu = fmaf(Grx, f.dx, Gwu);
    Gwv = fmaf(Gry, f.dy, Gwv);
    const float Gu = -Gpx, Gv = -Gpy;
    // camera Jacobian rows -> d0/d1 entries and the rotated point
    float Gd0[3], Gd1[3], Gxr0 = 0.f, Gxr1 = 0.f, Gxr2 = 0.f;
    if (DOF == 6) {
      const float ax = 2.f * f.xr0, ay = 2.f * f.xr1, az = 2.f * f.xr2;
      Gd0[0] = GJ0[0] - az * GJ0[4] + ay * GJ0[5];
      Gd0[1] = GJ0[1] + az * GJ0[3] - ax * GJ0[5];
      Gd0[2] = GJ0[2] - ay * GJ0[3] + ax * GJ0[4];
      Gd1[0] = GJ1[0] - az * GJ1[4] + ay * GJ1[5];
      Gd1[1] = GJ1[1] + az * GJ1[3] - ax * GJ1[5];
      Gd1[2] = GJ1[2] - ay * GJ1[3] + ax * GJ1[4];
      const float Gax = f.d0[2] * GJ0[4] - f.d0[1] * GJ0[5] + f.d1[2] * GJ1[4] - f.d1[1] * GJ1[5];
      const float Gay = -f.d0[2] * GJ0[3] + f.d0[0] * GJ0[5] - f.d1[2] * GJ1[3] + f.d1[0] * GJ1[5];
      const float Gaz = f.d0[1] * GJ0[3] - f.d0[0] * GJ0[4] + f.d1[1] * GJ1[3] - f.d1[0] * GJ1[4];
      Gxr0 = 2.f * Gax; Gxr1 = 2.f * Gay; Gxr2 = 2.f * Gaz;
    } else {
      Gd0[0] = GJ0[0] + f.xr2 * GJ0[3];
      Gd0[1] = GJ0[1];
      Gd0[2] = GJ0[2] - f.xr0 * GJ0[3];
      Gd1[0] = GJ1[0] + f.xr2 * GJ1[3];
      Gd1[1] = GJ1[1];
      Gd1[2] = GJ1[2] - f.xr0 * GJ1[3];
      Gxr2 = f.d0[0] * GJ0[3] + f.d1[0] * GJ1[3];
      Gxr0 = -(f.d0[2] * GJ0[3] + f.d1[2] * GJ1[3]);
    }
    float Grz = K[0] * Gd0[0] + K[1] * Gd0[1] + (K[2] - f.px) * Gd0[2] + K[3] * Gd1[0] + K[4] * Gd1[1] +
                (K[5] - f.py) * Gd1[2];
    Gpx = fmaf(-f.rz, Gd0[2], Gpx);
    Gpy = fmaf(-f.rz, Gd1[2], Gpy);
    if (BOUNDS) {   // the clamp passes no gradient where it is active
      if (f.ppx < bd.lbx || f.ppx > bd.ubx) Gpx = 0.f;
      if (f.ppy < bd.lby || f.ppy > bd.uby) Gpy = 0.f;
    }
    const float Ghx = Gpx * f.rz, Ghy = Gpy * f.rz;
    Grz = fmaf(Gpx, f.hx, fmaf(Gpy, f.hy, Grz));
    const float Gz = -f.rz * f.rz * Grz;
    const float Ghz = (f.hz >= p.z_min) ? Gz : 0.f;
    Gxr0 += K[0] * Ghx + K[3] * Ghy + K[6] * Ghz;
    Gxr1 += K[1] * Ghx + K[4] * Ghy + K[7] * Ghz;
    Gxr2 += K[2] * Ghx + K[5] * Ghy + K[8] * Ghz;
    const int n = n0 + tid;
    if (n < p.N) {
      const size_t i = (size_t)b * p.N + n;
      gx3d[i * 3 + 0] = R[0] * Gxr0 + R[3] * Gxr1 + R[6] * Gxr2;
      gx3d[i * 3 + 1] = R[1] * Gxr0 + R[4] * Gxr1 + R[7] * Gxr2;
      gx3d[i * 3 + 2] = R[2] * Gxr0 + R[5] * Gxr1 + R[8] * Gxr2;
      *reinterpret_cast<float2*>(gx2d + i * 2) = make_float2(Gu, Gv);
      *reinterpret_cast<float2*>(gw2d + i * 2) = make_float2(Gwu, Gwv);
    }
  }
  float one[1] = {gd};
  block_sum<1>(one, scratch);
  if (tid == 0) gdelta[b] = one[0];
}

// ------------------------------------------------------------------------------------------------------------
// fewest waves that keep the strided loop short (<= 8 points per lane up to 4 waves): as in the LM kernel, more waves
// per object only add reduction + barrier latency
static int gn_block_threads(int N) {
  int w = 1;
  while (w < 4 && 64 * 8 * w < N) w *= 2;
  return 64 * w;
}

int launch_gn_step_forward(const epropnp_problem* prob, float eps, const float* pose, float* step, float* pose_plus,
                           hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose || (!step && !pose_plus)) return fail(EPROPNP_EINVAL, "gn_step_forward: NULL pointer");
  const Problem d = to_device_problem(prob);
  const dim3 grid(padded_object_grid(d.B)), block(gn_block_threads(d.N));
  dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
    PNP_LAUNCH((gn_step_forward_kernel<decltype(DOF)::value, decltype(BND)::value>), grid, block,
               sizeof(float) * (block.x / 64) * kSumTStride<NormalEq<decltype(DOF)::value>::NV>, st, d, eps, pose, step, pose_plus);
    return 0;
  });
  return check_launch("gn_step_forward_kernel");
}

int launch_gn_step_backward(const epropnp_problem* prob, float eps, const float* pose, const float* grad_step,
                            const float* grad_pose_plus, float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose || (!grad_step && !grad_pose_plus) || !grad_x3d || !grad_x2d || !grad_w2d || !grad_delta)
    return fail(EPROPNP_EINVAL, "gn_step_backward: NULL pointer");
  const Problem d = to_device_problem(prob);
  const dim3 grid(padded_object_grid(d.B)), block(gn_block_threads(d.N));
  dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
    PNP_LAUNCH((gn_step_backward_kernel<decltype(DOF)::value, decltype(BND)::value>), grid, block,
               sizeof(float) * (block.x / 64) * kSumTStride<NormalEq<decltype(DOF)::value>::NV>, st, d, eps, pose, grad_step, grad_pose_plus, grad_x3d, grad_x2d, grad_w2d, grad_delta);
    return 0;
  });
  if (int rc = check_launch("gn_step_backward_kernel")) return rc;
  return launch_delta_path(prob, grad_delta, 1, grad_w2d, st);      // (no-op unless epropnp_problem.delta_stats is set)
}

}  // namespace pnp
