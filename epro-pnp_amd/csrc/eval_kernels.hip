// eval_kernels.hip -- the two "sweep" kernels of the EPro-PnP hot path for gfx950.
//
//   normal_equations_kernel : one pass over an object's N correspondences producing J^T J, J^T r and the
//       Huber cost at one pose.  HBM-bound: 28 B/point in, 4*(d(d+1)/2+d+1) B/object out; the (B,2N,d)
//       Jacobian the reference materialises (levenberg_marquardt.py:132) is never written.
//       Replaces evaluate_pnp(out_jacobian, out_residual, out_cost) + jac^T jac + jac^T r
//       (epropnp/common.py:67-100, levenberg_marquardt.py:205-214).
//   evaluate_cost_kernel : Huber cost of P poses per object (points loaded once, kept in registers).
//       Replaces evaluate_pnp(out_cost=True) with broadcast poses (common.py:67-100, camera.py:21-30).
#include "dispatch.h"
#include "pnp_host.h"

namespace pnp {

// ----------------------------------------------------------------------------------------------------------
template <int DOF, bool BOUNDS>
__global__ __launch_bounds__(1024) void normal_equations_kernel(Problem p, const float* __restrict__ pose, int clip,
                                                                  float* __restrict__ jtj, float* __restrict__ jtr,
                                                                  float* __restrict__ cost) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NH = NormalEq<DOF>::NH, NV = NormalEq<DOF>::NV;
  __shared__ float scratch[NV * 16];
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
  for (int n = (int)threadIdx.x; n < p.N; n += (int)blockDim.x) {
    const Point q = load_point(p, b, n);
    point_normal_eq<DOF, BOUNDS>(q, K, R, ps, p.z_min, delta, bd, clip != 0, acc);
  }
  block_sum<NV>(acc, scratch);
  if (threadIdx.x == 0) {
    int idx = 0;
#pragma unroll
    for (int i = 0; i < DOF; ++i)
#pragma unroll
      for (int j = i; j < DOF; ++j) {
        jtj[(size_t)b * DOF * DOF + i * DOF + j] = acc[idx];
        jtj[(size_t)b * DOF * DOF + j * DOF + i] = acc[idx];
        ++idx;
      }
#pragma unroll
    for (int i = 0; i < DOF; ++i) jtr[(size_t)b * DOF + i] = acc[NH + i];
    cost[b] = acc[NV - 1];
  }
}

// ----------------------------------------------------------------------------------------------------------
template <int DOF, int PPL, bool BOUNDS, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void evaluate_cost_kernel(Problem p, const float* __restrict__ poses, int P,
                                                                    float* __restrict__ cost) {
  constexpr int PL = PoseLen<DOF>::value;
  __shared__ float scratch[4 * 16];
  const int b = object_of_block(p.B);
  if (b >= p.B) return;
  float K[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, K, bd, delta);
  Point pts[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) pts[k] = load_point(p, b, (int)threadIdx.x + k * (int)blockDim.x);

  for (int j0 = 0; j0 < P; j0 += 4) {
    float c[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      c[jj] = 0.f;
      const int j = j0 + jj;
      if (j < P) {   // uniform branch
        float ps[PL], R[9], KR[9], Kt[3];
#pragma unroll
        for (int i = 0; i < PL; ++i) ps[i] = poses[((size_t)j * p.B + b) * PL + i];
        pose_to_rot<DOF>(ps, R);
        compose_kr_kt(K, R, ps, KR, Kt);
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          // exact Huber form on the IEEE path (matches cost_fun.py:8-12 bit-for-bit per point)
          const Point& q = pts[k];
          const float hx = KR[0] * q.X + KR[1] * q.Y + KR[2] * q.Z + Kt[0];
          const float hy = KR[3] * q.X + KR[4] * q.Y + KR[5] * q.Z + Kt[1];
          const float hz = KR[6] * q.X + KR[7] * q.Y + KR[8] * q.Z + Kt[2];
          const float z = fmaxf(hz, p.z_min);
          float px = hx / z, py = hy / z;
          if (BOUNDS) {
            px = fminf(fmaxf(px, bd.lbx), bd.ubx);
            py = fminf(fmaxf(py, bd.lby), bd.uby);
          }
          const float rx = (px - q.u) * q.wu, ry = (py - q.v) * q.wv;
          c[jj] += huber_exact(sqrtf(rx * rx + ry * ry), delta);
        }
      }
    }
    block_sum<4>(c, scratch);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (j0 + jj < P) cost[(size_t)(j0 + jj) * p.B + b] = c[jj];
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
int launch_normal_equations(const epropnp_problem* prob, const float* pose, int clip_jac, float* jtj, float* jtr,
                            float* cost, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose || !jtj || !jtr || !cost) return fail(EPROPNP_EINVAL, "normal_equations: NULL pointer");
  const Problem d = to_device_problem(prob);
  // streaming kernel: no register residency constraint; 1..4 waves per object
  int waves = 1;
  while (waves < 4 && 64 * 8 * waves < d.N) waves *= 2;
  while (waves < 4 && (long)d.B * waves < 8192) waves *= 2;
  const dim3 grid(padded_object_grid(d.B)), block(64 * waves);
  int rc = dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
    PNP_LAUNCH((normal_equations_kernel<decltype(DOF)::value, decltype(BND)::value>), grid, block, 0, st, d, pose,
               clip_jac, jtj, jtr, cost);
    return 0;
  });
  (void)rc;
  return check_launch("normal_equations_kernel");
}

int launch_evaluate_cost(const epropnp_problem* prob, const float* poses, int num_poses, float* cost, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0 || num_poses == 0) return EPROPNP_OK;
  if (!poses || !cost) return fail(EPROPNP_EINVAL, "evaluate_cost: NULL pointer");
  if (prob->num_pts > kMaxResidentPoints)
    return fail(EPROPNP_EINVAL, "evaluate_cost: num_pts %d exceeds the register-resident limit %d", prob->num_pts,
                kMaxResidentPoints);
  const Problem d = to_device_problem(prob);
  const Shape s = choose_shape(d.B, d.N);
  const dim3 grid(padded_object_grid(d.B)), block(64 * s.waves);
  dispatch_shape(prob->dof, s.ppl, has_bounds(prob), s.waves, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    PNP_LAUNCH((evaluate_cost_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value,
                                     decltype(MAXW)::value>),
               grid, block, 0, st, d, poses, num_poses, cost);
    return 0;
  });
  return check_launch("evaluate_cost_kernel");
}

}  // namespace pnp
