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
// SPLIT (few objects with many points: LineMOD's 32 crops x 4096 dense correspondences keep 32 of 256 CUs busy otherwise): G
// workgroups share an object's points.  Every part runs the whole iteration -- damping, solve, trust region are deterministic
// functions of the reduced normal equations, so the G copies stay identical -- but sweeps only its own points; after each
// sweep the parts' partial normal equations (NV floats each) meet in caller-provided scratch: relaxed agent-scope atomics,
// the data is its own arrival flag (the same exchange as the AMIS forward's split, csrc/amis_forward_mfma.hip).
template <int DOF, int PPL, bool BOUNDS, int MAXW, bool SPLIT = false>
__global__ __launch_bounds__(MAXW == 0 ? 256 : MAXW * 64) void lm_solve_kernel(Problem p, LmParams lm, const float* __restrict__ pose_init,
                                                               float* __restrict__ pose_opt, float* __restrict__ pose_cov,
                                                               float* __restrict__ cost_out, int* __restrict__ accept_out,
                                                               int nsplit, float* __restrict__ xch, StartSelect sel) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NH = NormalEq<DOF>::NH, NV = NormalEq<DOF>::NV;
  static_assert(!SPLIT || (MAXW > 0 && MAXW <= 4 && PPL > 0), "the split serves the register-resident <= 4-wave variants");
  // dynamic LDS: transposed reduction scratch (waves * kSumTStride<NV>) for <= 4 waves, NV * 16 for the DPP fallback
  // (+ 32 + 8 NV + 4 floats for the split's exchange)
  PNP_DYN_SMEM(float, scratch);
  constexpr bool kRow = (MAXW == 0);
  const int G = SPLIT ? nsplit : 1;
  int b_raw, part = 0;
  if (SPLIT) {           // all parts of an object on one XCD (workgroup g -> XCD g % 8)
    const int g = (int)blockIdx.x, per = (p.B + 7) >> 3, idx = g >> 3;
    part = idx % G;
    b_raw = (g & 7) * per + idx / G;
  } else {
    b_raw = kRow ? (int)(blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4)) : object_of_block(p.B);
  }
  if (!kRow && b_raw >= p.B) return;
  const bool live = b_raw < p.B;          // row variant: padded rows compute on object B-1 and write nothing
  const int b = live ? b_raw : p.B - 1;
  const bool writer = kRow ? (live && (threadIdx.x & 15u) == 0) : (threadIdx.x == 0 && part == 0);
  const int NS = lm.fast_mode ? (lm.num_iter > 0 ? lm.num_iter : 1) : lm.num_iter + 1;      // sweeps per solve
  int sweep_idx = 0;

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
    pts[k] = load_point(p, b, kRow ? (int)(threadIdx.x & 15u) : (SPLIT ? part * PPL * (int)blockDim.x : 0) + (int)threadIdx.x + k * (int)blockDim.x);

  float pose[PL];
  const float* start = pose_init + (size_t)b * PL;
  if (sel.cand != nullptr) {       // the split initialiser's winner (lm_core.h: StartSelect); wave-uniform except in the row variant
    int w = 0;
    float wc = sel.cand[(size_t)b * (PL + 1)];
    for (int q = 1; q < sel.parts; ++q) {
      const float c = sel.cand[((size_t)q * p.B + b) * (PL + 1)];
      if (c < wc) { wc = c; w = q; }
    }
    start = sel.cand + ((size_t)w * p.B + b) * (PL + 1) + 1;
    if (sel.rival_pose != nullptr && sel.rival_cost[b] < wc) start = sel.rival_pose + (size_t)b * PL;
  }
#pragma unroll
  for (int i = 0; i < PL; ++i) pose[i] = start[i];

  unsigned gone_parts = 0u;      // (SPLIT) sibling parts that have timed out once: later sweeps do not wait for them again
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
    if (SPLIT) {
      // this part's NV sums -> slot [b][sweep][part][NV] and xq[part]; lanes 0..NV-1 gather the siblings' rows (polled until
      // no longer the fill pattern, wave_ops.h: xwg_*).  A part that does not show up within the timeout (not resident: CU
      // mask, partitioned GPU, a foreign kernel holding CUs) is RECOMPUTED here from its points by the same lanes in the same
      // order -- the same bits -- so the result never depends on co-residency.  Then the G rows are added in part order and
      // the totals handed to everyone through LDS.
      const int tid = (int)threadIdx.x, T = (int)blockDim.x;
      float* xl = scratch + (T >> 6) * kSumTStride<NV>;        // [32] totals
      float* xq = xl + 32;                                     // [8][NV] the parts' rows
      unsigned* missw = reinterpret_cast<unsigned*>(xq + 8 * NV);
      unsigned* slot = reinterpret_cast<unsigned*>(xch) + (((size_t)b * NS + sweep_idx) * G) * NV;
      auto mine_of = [&](const float (&v)[NV]) {
        float m = v[0];
#pragma unroll
        for (int i = 1; i < NV; ++i) m = (tid == i) ? v[i] : m;
        return m;
      };
      if (tid == 0) *missw = 0u;
      __syncthreads();
      if (tid < NV) {
        const unsigned bits = xwg_payload(mine_of(acc));
        xwg_store(slot + part * NV + tid, bits);
        xq[part * NV + tid] = bits_f32(bits);
        unsigned miss = 0u;
        for (int q = 0; q < G; ++q) {
          if (q == part) continue;
          // (a part that was missing in an earlier sweep is not waited for again: whatever is there is taken, the rest recomputed)
          const unsigned u = xwg_poll(slot + q * NV + tid, ((gone_parts >> q) & 1u) ? 0u : lm.split_timeout);
          if (u == kXwgEmpty) miss |= 1u << q; else xq[q * NV + tid] = bits_f32(u);
        }
        if (miss) atomicOr(reinterpret_cast<int*>(missw), (int)miss);
      }
      __syncthreads();
      const unsigned todo = *missw;                 // the same in every thread
      gone_parts |= todo;
      if (todo) {
        if (tid == 0) raise_status(p, EPROPNP_ST_SPLIT_TIMEOUT, b);      // informational: slower, not wrong
        for (int q = 0; q < G; ++q) {
          if (!((todo >> q) & 1u)) continue;
          float a2[NV];
#pragma unroll
          for (int i = 0; i < NV; ++i) a2[i] = 0.f;
#pragma unroll
          for (int k = 0; k < PPL; ++k)
            point_normal_eq<DOF, BOUNDS>(load_point(p, b, q * PPL * T + tid + k * T), K, R, t, z_min, delta, inv_eps, bd, clip, a2);
          block_sum_t<NV>(a2, scratch);
          if (tid < NV) xq[q * NV + tid] = bits_f32(xwg_payload(mine_of(a2)));
          __syncthreads();
        }
      }
      if (tid < NV) {
        float tot = 0.f;
        for (int q = 0; q < G; ++q) tot += xq[q * NV + tid];
        xl[tid] = tot;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] = xl[i];
      __syncthreads();
      ++sweep_idx;
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

// Parts per object for the split, measured (profiles/r03_lm_split.txt): an exchange costs ~3.5 us per sweep, which only pays
// when an object's sweep keeps a whole CU busy for longer -- beyond 2048 points (32 x 4096, LM 5: 97.8 -> 59.6 us; 64 x 2048:
// 51.5 -> 55.6, 32 x 512: 28.9 -> 50).  Then: the most parts (<= 8) with at least 256 points each that keep the grid at one
// workgroup per CU (the parts of an object wait for each other).  EPROPNP_LM_SPLIT=<G> overrides (1: off).
static int lm_split_parts(int B, int N) {
  const long wgs = padded_object_grid(B), cus = device_cu_count();
  int g = (N > 2048) ? 8 : 1;
  while (g > 1 && (wgs * g > cus || (long)N < 256L * g)) g >>= 1;
  { int ov[1]; if (env_ints("EPROPNP_LM_SPLIT", ov, 1) && (ov[0] == 1 || ov[0] == 2 || ov[0] == 4 || ov[0] == 8) && wgs * ov[0] <= 4096 && (long)N >= 64L * ov[0]) g = ov[0]; }
  return g;
}

static int lm_sweeps(const epropnp_lm_params* lm) {
  return lm->fast_mode ? (lm->num_iter > 0 ? lm->num_iter : 1) : lm->num_iter + 1;
}

unsigned long long lm_split_bytes(const epropnp_problem* prob, const epropnp_lm_params* lm) {
  if (prob == nullptr || lm == nullptr || prob->num_obj <= 0 || prob->num_pts <= 16 || prob->num_pts > kMaxResidentPoints) return 0;
  const int g = lm_split_parts(prob->num_obj, prob->num_pts);
  const int NV = prob->dof == 6 ? NormalEq<6>::NV : NormalEq<4>::NV;
  if (g <= 1 || (prob->num_pts + g - 1) / g > 1024) return 0;       // a part: 4 waves x <= 4 points per lane
  return sizeof(float) * (unsigned long long)prob->num_obj * lm_sweeps(lm) * g * NV;
}

int launch_lm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, const float* pose_init, float* pose_opt,
                    float* pose_cov, float* cost, int32_t* accept_mask, void* split_scratch,
                    unsigned long long split_scratch_bytes, hipStream_t st, const StartSelect* select) {
  if (int rc = check_problem(prob)) return rc;
  if (!lm) return fail(EPROPNP_EINVAL, "lm_solve: params NULL");
  if (prob->num_obj == 0) return EPROPNP_OK;
  StartSelect sel;
  sel.cand = nullptr; sel.rival_pose = nullptr; sel.rival_cost = nullptr; sel.parts = 0;
  if (select != nullptr && select->cand != nullptr) sel = *select;
  if ((!pose_init && sel.cand == nullptr) || !pose_opt) return fail(EPROPNP_EINVAL, "lm_solve: NULL pose pointer");
  if (lm->num_iter < 0 || lm->num_iter > 31 * 1000) return fail(EPROPNP_EINVAL, "lm_solve: bad num_iter");
  const Problem d = to_device_problem(prob);
  LmParams k;
  k.num_iter = lm->num_iter; k.fast_mode = lm->fast_mode;
  k.min_diag = lm->min_lm_diagonal; k.max_diag = lm->max_lm_diagonal;
  k.min_rel_decrease = lm->min_relative_decrease; k.radius0 = lm->initial_trust_region_radius;
  k.radius_max = lm->max_trust_region_radius; k.eps = lm->eps; k.split_timeout = split_timeout_cycles();
  if (d.N <= 16 && !tune_flag("lm_no_rows")) {   // RSLM sub-problems: 16 objects per 256-thread block
    const dim3 grid((d.B + 15) / 16), block(256);
    dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
      PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, 1, decltype(BND)::value, 0>), grid, block, 0, st, d, k, pose_init,
                 pose_opt, pose_cov, cost, accept_mask, 1, (float*)nullptr, sel);
      return 0;
    });
    return check_launch("lm_solve_kernel (row variant)");
  }
  if (d.N > kMaxResidentPoints) {   // streaming: 8 waves per object, points re-read on every sweep
    const dim3 grid(padded_object_grid(d.B)), block(512);
    dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
      PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, 0, decltype(BND)::value, 8>), grid, block,
                 sizeof(float) * NormalEq<decltype(DOF)::value>::NV * 16, st, d, k, pose_init, pose_opt, pose_cov, cost,
                 accept_mask, 1, (float*)nullptr, sel);
      return 0;
    });
    return check_launch("lm_solve_kernel (streaming)");
  }
  {   // few objects with many points: G workgroups per object (kernel comment); 4 waves per part, <= 2 points per lane
    const int G = lm_split_parts(d.B, d.N);
    const int NVh = prob->dof == 6 ? NormalEq<6>::NV : NormalEq<4>::NV;
    const size_t need = sizeof(float) * (size_t)d.B * lm_sweeps(lm) * G * NVh;
    const int per_part = (d.N + G - 1) / G;
    if (G > 1 && split_scratch != nullptr && split_scratch_bytes >= need && per_part <= 1024) {
      const int ppl = per_part > 512 ? 4 : (per_part > 256 ? 2 : 1);
      if (!exchange_prefilled(split_scratch, need) &&
          launch_fill_u32(split_scratch, 0xffffffffu, need / 4, st) != EPROPNP_OK) {      // (a kernel, not a memset node)
        (void)hipGetLastError();
        return fail(EPROPNP_ELAUNCH, "lm_solve: could not fill the split scratch");
      }
      const dim3 grid(padded_object_grid(d.B) * G), block(256);
      dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
        constexpr int NVc = NormalEq<decltype(DOF)::value>::NV;
        const size_t smem = sizeof(float) * (4 * kSumTStride<NVc> + 32 + 8 * NVc + 4);
        if (ppl == 4) {
          PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, 4, decltype(BND)::value, 4, true>), grid, block, smem, st, d, k,
                     pose_init, pose_opt, pose_cov, cost, accept_mask, G, (float*)split_scratch, sel);
        } else if (ppl == 2) {
          PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, 2, decltype(BND)::value, 4, true>), grid, block, smem, st, d, k,
                     pose_init, pose_opt, pose_cov, cost, accept_mask, G, (float*)split_scratch, sel);
        } else {
          PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, 1, decltype(BND)::value, 4, true>), grid, block, smem, st, d, k,
                     pose_init, pose_opt, pose_cov, cost, accept_mask, G, (float*)split_scratch, sel);
        }
        return 0;
      });
      return check_launch("lm_solve_kernel (split over workgroups)");
    }
  }
  // fewest waves per object at every batch size: more waves only add cross-wave reduction + barrier latency to each of
  // the 1+L dependent sweeps (measured on MI355X at B = 32 / 256 / 600: 1 wave 38 / 30 / 30 us vs 86 / 62 / 41 us)
  Shape s = choose_shape(d.B, d.N, /*max_ppl=*/8, /*want_waves_total=*/0);
  int ov[2];
  if (tune_ints("lm_shape", ov, 2) && valid_shape_override(ov[0], ov[1], d.N)) { s.waves = ov[0]; s.ppl = ov[1]; }
  const dim3 grid(padded_object_grid(d.B)), block(64 * s.waves);
  dispatch_shape(prob->dof, s.ppl, has_bounds(prob), s.waves, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    PNP_LAUNCH((lm_solve_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value, decltype(MAXW)::value>),
               grid, block,
               sizeof(float) * (decltype(MAXW)::value <= 4 ? s.waves * kSumTStride<NormalEq<decltype(DOF)::value>::NV>
                                                           : NormalEq<decltype(DOF)::value>::NV * 16),
               st, d, k, pose_init, pose_opt, pose_cov, cost, accept_mask, 1, (float*)nullptr, sel);
    return 0;
  });
  return check_launch("lm_solve_kernel");
}

}  // namespace pnp
