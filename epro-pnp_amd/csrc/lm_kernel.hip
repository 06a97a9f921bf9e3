// lm_kernel.hip -- the whole Levenberg-Marquardt / Gauss-Newton PnP solve in ONE kernel for gfx950.
//
// Replaces LMSolver.solve with a given starting pose (epropnp/levenberg_marquardt.py:132-190), _lm_iter
// (:192-241) and pose_add (:255-265).  The reference launches ~40 ATen kernels per iteration and keeps two
// (B,2N,d) Jacobians in memory; here one workgroup owns one object, its N correspondences are read from HBM
// once and stay in registers for all 1+L sweeps, J^T J / J^T r / cost are wave-reduced with DPP, and the
// d x d damped system is solved in registers by a Jacobi-scaled fp32 Cholesky (the matrix is SPD by construction;
// the reference's unscaled LU with pivoting gives the same solution up to its own, larger, rounding error).
#include "dispatch.h"
#include "lm_core.h"
#include "pnp_host.h"

namespace pnp {

// MAXW == 0 selects the small-problem variant: N <= 16 points, one object per 16-lane DPP row (4 objects per wave,
// reductions are row_ror adds only) -- the shape of the RSLM initialiser's 10^4..10^5 sub-problems.
// PPL == 0 selects the streaming variant for N beyond the register-resident limit (8192): every sweep re-reads the
// object's points from HBM / L2 (the reference accepts any N).
template <int DOF, int PPL, bool BOUNDS, int MAXW>
__global__ __launch_bounds__(MAXW == 0 ? 256 : MAXW * 64) void lm_solve_kernel(Problem p, LmParams lm, const float* __restrict__ pose_init,
                                                               float* __restrict__ pose_opt, float* __restrict__ pose_cov,
                                                               float* __restrict__ cost_out, int* __restrict__ accept_out) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NH = NormalEq<DOF>::NH, NV = NormalEq<DOF>::NV;
  // dynamic LDS: transposed reduction scratch (waves * kSumTStride<NV>) for <= 4 waves, NV * 16 for the DPP fallback
  PNP_DYN_SMEM(float, scratch);
  constexpr bool kRow = (MAXW == 0);
  const int b_raw = kRow ? (int)(blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4)) : object_of_block(p.B);
  if (!kRow && b_raw >= p.B) return;
  const bool live = b_raw < p.B;          // row variant: padded rows compute on object B-1 and write nothing
  const int b = live ? b_raw : p.B - 1;
  const bool writer = kRow ? (live && (threadIdx.x & 15u) == 0) : (threadIdx.x == 0);

  float K[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, K, bd, delta);
  // wave-uniform sweep operands go to VGPRs (an SGPR source operand halves the VALU issue rate on gfx950)
#pragma unroll
  for (int i = 0; i < 9; ++i) K[i] = to_vgpr(K[i]);
  delta = to_vgpr(delta);
  const float z_min = to_vgpr(p.z_min), inv_eps = to_vgpr(p.inv_huber_eps);
  Point pts[PPL > 0 ? PPL : 1];
#pragma unroll
  for (int k = 0; k < PPL; ++k)
    pts[k] = load_point(p, b, kRow ? (int)(threadIdx.x & 15u) : (int)threadIdx.x + k * (int)blockDim.x);

  float pose[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) pose[i] = pose_init[(size_t)b * PL + i];

  // one sweep: normal equations + cost of all points at pose `ps`
  auto sweep = [&](const float* ps, bool clip, float (&acc)[NV]) {
    float R[9], t[3];
    pose_to_rot<DOF>(ps, R);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = to_vgpr(R[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = to_vgpr(ps[i]);
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
#pragma unroll
    for (int k = 0; k < PPL; ++k) point_normal_eq<DOF, BOUNDS>(pts[k], K, R, t, z_min, delta, inv_eps, bd, clip, acc);
    if (PPL == 0) {
      for (int n = (int)threadIdx.x; n < p.N; n += (int)blockDim.x)
        point_normal_eq<DOF, BOUNDS>(load_point(p, b, n), K, R, t, z_min, delta, inv_eps, bd, clip, acc);
    }
    if (kRow) {
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] = row_sum16(acc[i]);
    } else if (MAXW <= 4) {
      block_sum_t<NV>(acc, scratch);
    } else {
      block_sum<NV>(acc, scratch);
    }
  };

  float cur[NV];
  int accepted_bits = 0, st_bits = 0;
  lm_iterate<DOF>(lm, sweep, pose, cur, accepted_bits, st_bits);
#pragma unroll
  for (int i = 0; i < PL; ++i) st_bits |= isfinite(pose[i]) ? 0 : EPROPNP_ST_NONFINITE_POSE;
  if (writer) raise_status(p, st_bits, b);

  if (writer) {
#pragma unroll
    for (int i = 0; i < PL; ++i) pose_opt[(size_t)b * PL + i] = pose[i];
    if (cost_out) cost_out[b] = cur[NV - 1];
    if (accept_out) accept_out[b] = accepted_bits;
  }
  if (pose_cov) {   // inverse(J^T J + eps I) at the final accepted point (:170-181)
    float H[DOF][DOF], Hi[DOF][DOF];
    ScaledFactor<DOF> f;
    unpack_h<DOF>(cur, H);
#pragma unroll
    for (int i = 0; i < DOF; ++i) H[i][i] += lm.eps;
    scaled_cholesky<DOF>(H, f);
    scaled_inverse<DOF>(f, Hi);
    if (writer) {
#pragma unroll
      for (int i = 0; i < DOF; ++i)
#pragma unroll
        for (int j = 0; j < DOF; ++j) pose_cov[(size_t)b * DOF * DOF + i * DOF + j] = Hi[i][j];
    }
  }
}

int launch_lm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, const float* pose_init, float* pose_opt,
                    float* pose_cov, float* cost, int32_t* accept_mask, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (!lm) return fail(EPROPNP_EINVAL, "lm_solve: params NULL");
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose_init || !pose_opt) return fail(EPROPNP_EINVAL, "lm_solve: NULL pose pointer");
  if (lm->num_iter < 0 || lm->num_iter > 31 * 1000) return fail(EPROPNP_EINVAL, "lm_solve: bad num_iter");
  const Problem d = to_device_problem(prob);
  LmParams k;
  k.num_iter = lm->num_iter; k.fast_mode = lm->fast_mode;
  k.min_diag = lm->min_lm_diagonal; k.max_diag = lm->max_lm_diagonal;
  k.min_rel_decrease = lm->min_relative_decrease; k.radius0 = lm->initial_trust_region_radius;
  k.radius_max = lm->max_trust_region_radius; k.eps = lm->eps;
  if (d.N <= 16 && !getenv("EPROPNP_LM_NO_ROWS")) {   // RSLM sub-problems: 16 objects per 256-thread block
    const dim3 grid((d.B + 15) / 16), block(256);
    dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
      PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, 1, decltype(BND)::value, 0>), grid, block, 0, st, d, k, pose_init,
                 pose_opt, pose_cov, cost, accept_mask);
      return 0;
    });
    return check_launch("lm_solve_kernel (row variant)");
  }
  if (d.N > kMaxResidentPoints) {   // streaming: 8 waves per object, points re-read on every sweep
    const dim3 grid(padded_object_grid(d.B)), block(512);
    dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
      PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, 0, decltype(BND)::value, 8>), grid, block,
                 sizeof(float) * NormalEq<decltype(DOF)::value>::NV * 16, st, d, k, pose_init, pose_opt, pose_cov, cost,
                 accept_mask);
      return 0;
    });
    return check_launch("lm_solve_kernel (streaming)");
  }
  // fewest waves per object at every batch size: more waves only add cross-wave reduction + barrier latency to each of
  // the 1+L dependent sweeps (measured on MI355X at B = 32 / 256 / 600: 1 wave 38 / 30 / 30 us vs 86 / 62 / 41 us)
  Shape s = choose_shape(d.B, d.N, /*max_ppl=*/8, /*want_waves_total=*/0);
  int ov[2];
  if (env_ints("EPROPNP_LM_SHAPE", ov, 2) && valid_shape_override(ov[0], ov[1], d.N)) { s.waves = ov[0]; s.ppl = ov[1]; }
  const dim3 grid(padded_object_grid(d.B)), block(64 * s.waves);
  dispatch_shape(prob->dof, s.ppl, has_bounds(prob), s.waves, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value, decltype(MAXW)::value>),
               grid, block,
               sizeof(float) * (decltype(MAXW)::value <= 4 ? s.waves * kSumTStride<NormalEq<decltype(DOF)::value>::NV>
                                                           : NormalEq<decltype(DOF)::value>::NV * 16),
               st, d, k, pose_init, pose_opt, pose_cov, cost, accept_mask);
    return 0;
  });
  return check_launch("lm_solve_kernel");
}

}  // namespace pnp
