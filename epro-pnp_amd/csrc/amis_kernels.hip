// amis_kernels.hip -- the AMIS Monte-Carlo pose sampler (forward) and its gradient (backward) for gfx950: the all-VALU
// kernels plus the two launchers.  The default paths are the matrix-core variants (amis_forward_mfma.hip,
// amis_backward_mfma.hip), which share the sampler stages of amis_common.h; the kernels in this file are selected with
// EPROPNP_{FWD,BWD}_IMPL=valu and take over where the MFMA backward's LDS pose table does not fit (> ~2700 samples).
//
// Forward replaces the loop of EProPnPBase.monte_carlo_forward (epropnp/epropnp.py:132-182) with ONE kernel:
// initial_fit (:216-220,:288-302), sampling from Student-t x {ACG | von-Mises/uniform mix}
// (pyro MultivariateStudentT, epropnp/distributions.py:42-72), the cost sweep over s poses x N points (:151),
// proposal log-densities and the mixture weight algebra (:156-169), and estimate_params (:238-260,:317-342).
// The reference materialises (s,B,N,3) broadcasts and keeps them for autograd (~12 MB per object); here one
// workgroup owns one object, its points sit in registers ("lane = point" during the sweep), samples live in LDS
// ("lane = sample" for everything else) and the two views are bridged with v_readlane broadcasts.
//
// Backward replaces autograd's replay of the same sweep: gradients of
//   sum_j g_logw[j] * (-cost(pose_j)) + g_init * cost(pose_init)
// w.r.t. x3d, x2d, w2d, delta are RECOMPUTED from the points (lane = point, loop over poses; no atomics,
// nothing saved by the forward except the pose samples themselves).
#include "dispatch.h"
#include "amis_common.h"

namespace pnp {

// ================================================================================================================
// forward
// ================================================================================================================
template <int DOF, int PPL, bool BOUNDS, int MAXW>
__global__ __launch_bounds__(MAXW * 64, (MAXW == 4 ? (PPL >= 8 ? 2 : 3) : 1)) void amis_forward_kernel(Problem p, AmisParams a_in,
                                                                   const float* __restrict__ pose_opt,
                                                                   const float* __restrict__ pose_cov,
                                                                   const float* __restrict__ noise,
                                                                   float* __restrict__ pose_samples,
                                                                   float* __restrict__ logweights,
                                                                   float* __restrict__ proposals) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NZ = (DOF == 6) ? 8 : 4 + 3 * kVmTries;
  const int b = object_of_block(p.B);
  if (b >= p.B) return;
  AmisParams a = a_in;
  if (a.offset_dev != nullptr) a.offset += *a.offset_dev;
  const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id();
  const int S = a.S, K = a.K, s = S / K, WP = a.WP, WS = (T >> 6) / WP;
  const int wp = wv % WP, ws = wv / WP;

  PNP_DYN_SMEM(float, smem);
  float* ptab = smem;                 // [s][12] K R | K t of the current iteration's samples (16-B aligned rows)
  float* wred = ptab + 12 * s;        // [waves][kWaveRed] transposed cost reduction (float4 views: 16-B aligned)
  float* smp = wred + (T >> 6) * kWaveRed;   // [PL][S]
  float* cst = smp + PL * S;          // [S]   cost of each sample
  float* mixl = cst + S;              // [S]   log sum_j q_j(sample)
  float* lgw = mixl + S;              // [S]   log weight
  float* cpart = lgw + S;             // [WP][s] partial costs of the current iteration
  float* prop = cpart + WP * s;       // [K][kPropStride]
  float* red = prop + K * kPropStride;   // [16*16] reduction scratch

  float Kc[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, Kc, bd, delta);
  // this wave's slice of the points: lane = point; kept in the pre-multiplied form the sweep consumes
  SweepPoint pts[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) pts[k] = to_sweep_point(load_point(p, b, wp * 64 + lane + k * 64 * WP));
  // wave-uniform operands of the sweep live in VGPRs: an SGPR source halves the VALU issue rate on gfx950
  // (profiles/r01_ubench_valu_rates.txt: v_fma_f32 1.05 ns vs 1.84 ns per wave-instruction with an SGPR operand)
  const float zmin_v = to_vgpr(p.z_min), delta_v = to_vgpr(delta);

  if (tid < (DOF == 6 ? 2 : 1))
    initial_fit<DOF>(pose_opt + (size_t)b * PL, pose_cov + (size_t)b * DOF * DOF, a.eps, a.dispersion, prop, tid);
  if (tid == (int)blockDim.x - 1) denormalise_pose_opt<DOF>(a, pose_opt, b);
  __syncthreads();

  AmisCtx cx;
  cx.ptab = ptab; cx.smp = smp; cx.cst = cst; cx.mixl = mixl; cx.lgw = lgw; cx.cpart = cpart; cx.prop = prop; cx.red = red;
  cx.S = S; cx.K = K; cx.s = s; cx.T = T; cx.tid = tid; cx.b = b; cx.cstride = s;
  cx.nzb = nullptr;
  cx.rred = nullptr;      // (this kernel's LDS budget is its sample table: the refit sums stay on DPP / readlane chains)

  for (int it = 0; it < K; ++it) {
    amis_draw<DOF>(cx, p, a, it, Kc, noise, pose_samples);
    __syncthreads();

    // ---------------- 2. cost sweep: s poses x this wave's points (lane = point) ----------------
    const int ntile = (s + 63) >> 6;
    if (PNP_ABLATED(a, 1)) {
      for (int n = tid; n < s; n += T) cpart[n] = 1.0f;
    } else
    for (int t = ws; t < ntile; t += WS) {
      const int base = t * 64;
      const int cnt = min(64, s - base);
      float mine = 0.f;
      // Per-lane partial costs of 16 samples are parked in LDS (one ds_write each, no dependent chain in the hot
      // loop) and summed "transposed": lane l adds the 16 partials of sample (l & 15) held by lanes 16q..16q+15
      // (q = l >> 4), then the four quarter sums are combined through a second 64-float exchange.
      float* rt = wred + wv * kWaveRed;
      float* rq = rt + 16 * kRedStride;
      for (int j0 = 0; j0 < cnt; j0 += 16) {
        const int g = min(16, cnt - j0);
        for (int jj = 0; jj < g; ++jj) {
          const float4* row = reinterpret_cast<const float4*>(ptab + 12 * (base + j0 + jj));   // uniform: LDS broadcast
          const float4 r0 = row[0], r1 = row[1], r2 = row[2];
          const float kr[9] = {r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, r2.x, r2.y, r2.z};
          const float kt[3] = {r0.w, r1.w, r2.w};
          float c = 0.f;
#pragma unroll
          for (int k = 0; k < PPL; ++k) c += sweep_cost<BOUNDS>(pts[k], kr, kt, zmin_v, delta_v, bd);
          rt[jj * kRedStride + lane] = c;
        }
        wave_lds_fence();
        const float4* col = reinterpret_cast<const float4*>(rt + (lane & 15) * kRedStride + 16 * (lane >> 4));
        const float4 v0 = col[0], v1 = col[1], v2 = col[2], v3 = col[3];
        const float quarter = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w)) +
                              (((v2.x + v2.y) + (v2.z + v2.w)) + ((v3.x + v3.y) + (v3.z + v3.w)));
        rq[(lane & 15) * 4 + (lane >> 4)] = quarter;
        wave_lds_fence();
        const float4 q4 = *reinterpret_cast<const float4*>(rq + (lane & 15) * 4);
        const float total = (q4.x + q4.y) + (q4.z + q4.w);
        mine = ((lane >> 4) == (j0 >> 4)) ? total : mine;   // lanes j0..j0+15 own samples j0..j0+15 of the tile
        wave_lds_fence();                                   // rt / rq are rewritten by the next group
      }
      if (base + lane < s) cpart[wp * s + base + lane] = mine;
    }
    __syncthreads();

    amis_weights<DOF>(cx, a, it, WP);
    __syncthreads();
    if (it == K - 1) break;
    amis_refit<DOF>(cx, a, it);
  }

  // ---------------- outputs ----------------
  {   // numerical events for the caller's status word (include/epropnp_hip.h); no-op without one
    int st_bits = 0;
    for (int m = tid; m < S; m += T) {
      const float lw = lgw[m];
      logweights[(size_t)m * p.B + b] = lw;
      st_bits |= (lw == lw && lw != INFINITY) ? 0 : EPROPNP_ST_NONFINITE_WEIGHT;     // -inf = zero weight is legitimate
    }
    for (int i = tid; i < K; i += T)
      st_bits |= (prop[i * kPropStride + 37] != 0.f || prop[i * kPropStride + 38] != 0.f) ? EPROPNP_ST_CHOL_FALLBACK : 0;
    raise_status(p, st_bits, b);
  }
  if (proposals != nullptr)
    for (int i = tid; i < K * kPropStride; i += T) proposals[(size_t)b * K * kPropStride + i] = prop[i];
  advance_counters(a, p.B);
}

// ================================================================================================================
// backward
// ================================================================================================================
template <int DOF, int PPL, bool BOUNDS, int MAXW>
__global__ __launch_bounds__(MAXW * 64, (MAXW == 4 ? (PPL >= 8 ? 2 : (PPL == 4 ? 3 : 4)) : 1)) void amis_backward_kernel(Problem p, const float* __restrict__ pose_samples,
                                                                    const float* __restrict__ g_logw, int S,
                                                                    const float* __restrict__ pose_init,
                                                                    const float* __restrict__ g_init,
                                                                    float* __restrict__ gx3d, float* __restrict__ gx2d,
                                                                    float* __restrict__ gw2d, float* __restrict__ gdelta,
                                                                    float drop_eps) {
  constexpr int PL = PoseLen<DOF>::value;
  // pose tiles: 64 poses x {K R (9) | K t (3) | weight | pad} as 4 float4 rows, double buffered
  __shared__ __attribute__((aligned(16))) float tab[2][64][16];
  __shared__ float red[16];
  __shared__ __attribute__((aligned(8))) float hist[kDropHistFloats];
  const int b = object_of_block(p.B);
  if (b >= p.B) return;
  const int T = (int)blockDim.x, tid = (int)threadIdx.x;

  float Kc[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, Kc, bd, delta);
  const float zmin_v = to_vgpr(p.z_min), delta_v = to_vgpr(delta);
  Point pts[PPL];
  float gX[PPL], gY[PPL], gZ[PPL], gu[PPL], gv[PPL], gwu[PPL], gwv[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    pts[k] = load_point(p, b, tid + k * T);
    gX[k] = gY[k] = gZ[k] = gu[k] = gv[k] = gwu[k] = gwv[k] = 0.f;
  }
  float gd = 0.f;

  const bool with_init = (pose_init != nullptr) && (g_init != nullptr);
  const int P = S + (with_init ? 1 : 0);      // pose index S = pose_init
  const int ntile = (P + 63) >> 6;
  // wave-uniform, hence a scalar load: fetched under a uniform branch, not inside a per-lane conditional (scalar loads
  // ignore EXEC; a NULL g_init would fault there even when no lane takes the arm)
  const float g_init_b = with_init ? g_init[b] : 0.f;

  // Samples whose total |weight| is below drop_eps of the object's total are skipped (mass_drop_threshold,
  // amis_common.h): at the default 2^-24 that is under the fp32 rounding of the S-term sums; ~17 % of the AMIS samples
  // after softmax normalisation (the Student-t tails).  drop_eps = 0: exact.
  float amax = 0.f;
  for (int m = tid; m < S; m += T) amax = fmaxf(amax, fabsf(g_logw[(size_t)m * p.B + b]));
  amax = block_max(amax, red);
  const float askip = mass_drop_threshold([&](int m) { return fabsf(g_logw[(size_t)m * p.B + b]); }, S, amax, drop_eps, hist);

  // lanes 0..63 of the workgroup fetch one pose each of tile t (global loads issued early, consumed late)
  float nps[PL], naw = 0.f;
  auto fetch = [&](int t) {
    if (tid < 64) {
      const int m = min(t * 64 + tid, P - 1);
      const float* src = (m < S) ? pose_samples + ((size_t)m * p.B + b) * PL : pose_init + (size_t)b * PL;
#pragma unroll
      for (int i = 0; i < PL; ++i) nps[i] = src[i];
      naw = (m < S) ? -g_logw[(size_t)m * p.B + b] : g_init_b;      // logw = -cost - const
      if (m < S && fabsf(naw) <= askip) naw = 0.f;
    }
  };
  auto publish = [&](int buf) {
    if (tid < 64) {
      float R[9], KR[9], Kt[3];
      pose_to_rot<DOF>(nps, R);
      compose_kr_kt(Kc, R, nps, KR, Kt);
      float4* row = reinterpret_cast<float4*>(&tab[buf][tid][0]);
      row[0] = make_float4(KR[0], KR[1], KR[2], KR[3]);
      row[1] = make_float4(KR[4], KR[5], KR[6], KR[7]);
      row[2] = make_float4(KR[8], Kt[0], Kt[1], Kt[2]);
      row[3] = make_float4(naw, 0.f, 0.f, 0.f);
    }
  };

  if (ntile > 0) {
    fetch(0);
    publish(0);
  }
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const bool more = (t + 1 < ntile);
    if (more) fetch(t + 1);
    const int cnt = min(64, P - t * 64);
    const int buf = t & 1;
    const float4* row0 = reinterpret_cast<const float4*>(&tab[buf][0][0]);   // uniform address: LDS broadcast
    float4 n0 = row0[0], n1 = row0[1], n2 = row0[2], n3 = row0[3];
    for (int j = 0; j < cnt; ++j) {
      const float4 r0 = n0, r1 = n1, r2 = n2, r3 = n3;
      {   // fetch the next pose row now; it is consumed one trip later
        const float4* rn = reinterpret_cast<const float4*>(&tab[buf][min(j + 1, cnt - 1)][0]);
        n0 = rn[0]; n1 = rn[1]; n2 = rn[2]; n3 = rn[3];
      }
      const float a = r3.x;
      if (uniform_is_zero(a)) continue;       // exact zero weight (wave-uniform)
      const float kr[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x};
      const float kt[3] = {r2.y, r2.z, r2.w};
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const Point& q = pts[k];
        const float hx = fmaf(kr[0], q.X, fmaf(kr[1], q.Y, fmaf(kr[2], q.Z, kt[0])));
        const float hy = fmaf(kr[3], q.X, fmaf(kr[4], q.Y, fmaf(kr[5], q.Z, kt[1])));
        const float hz = fmaf(kr[6], q.X, fmaf(kr[7], q.Y, fmaf(kr[8], q.Z, kt[2])));
        const float rz = fast_rcp(fmaxf(hz, zmin_v));
        const float ppx = hx * rz, ppy = hy * rz;          // un-clamped projection
        float px = ppx, py = ppy;
        if (BOUNDS) {
          px = clamp_lu(px, bd.lbx, bd.ubx);
          py = clamp_lu(py, bd.lby, bd.uby);
        }
        const float dx = px - q.u, dy = py - q.v;
        const float rx = dx * q.wu, ry = dy * q.wv;
        const float s2 = fmaf(rx, rx, ry * ry);
        const float rs = fast_rsqrt(fmaxf(s2, 1e-30f));
        const float rho = s2 * rs;
        const float mm = fminf(rho, delta_v);
        const float coef = a * mm * rs;                      // a * min(1, delta / rho)
        gd = fmaf(a, rho - mm, gd);                          // d huber / d delta = max(rho - delta, 0)
        const float crx = coef * rx, cry = coef * ry;
        gwu[k] = fmaf(crx, dx, gwu[k]);
        gwv[k] = fmaf(cry, dy, gwv[k]);
        float gpx = crx * q.wu, gpy = cry * q.wv;
        gu[k] += gpx;                                        // d/du = -g_p ; sign applied at the end
        gv[k] += gpy;
        if (BOUNDS) {                                        // clamp passes no gradient where it is active
          gpx = (ppx < bd.lbx || ppx > bd.ubx) ? 0.f : gpx;
          gpy = (ppy < bd.lby || ppy > bd.uby) ? 0.f : gpy;
        }
        const float ghx = gpx * rz, ghy = gpy * rz;
        float ghz = -(gpx * ppx + gpy * ppy) * rz;
        ghz = (hz >= zmin_v) ? ghz : 0.f;
        gX[k] = fmaf(kr[0], ghx, fmaf(kr[3], ghy, fmaf(kr[6], ghz, gX[k])));
        gY[k] = fmaf(kr[1], ghx, fmaf(kr[4], ghy, fmaf(kr[7], ghz, gY[k])));
        gZ[k] = fmaf(kr[2], ghx, fmaf(kr[5], ghy, fmaf(kr[8], ghz, gZ[k])));
      }
    }
    if (more) publish((t + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int n = tid + k * T;
    if (n < p.N) {
      const size_t i = (size_t)b * p.N + n;
      gx3d[i * 3] = gX[k]; gx3d[i * 3 + 1] = gY[k]; gx3d[i * 3 + 2] = gZ[k];
      *reinterpret_cast<float2*>(gx2d + i * 2) = make_float2(-gu[k], -gv[k]);
      *reinterpret_cast<float2*>(gw2d + i * 2) = make_float2(gwu[k], gwv[k]);
    }
  }
  float one[1] = {gd};
  block_sum<1>(one, red);
  if (tid == 0) gdelta[b] = one[0];
}

// ================================================================================================================
// launchers
// ================================================================================================================
int launch_amis_forward(const epropnp_problem* prob, const epropnp_amis_params* am, const float* pose_opt,
                        const float* pose_cov, const float* noise, float* pose_samples, float* logweights,
                        float* proposals, hipStream_t st, const DenormOut* dn) {
  if (int rc = check_problem(prob)) return rc;
  if (!am) return fail(EPROPNP_EINVAL, "amis_forward: params NULL");
  if (am->num_iter <= 0 || am->mc_samples <= 0 || am->mc_samples % am->num_iter != 0)
    return fail(EPROPNP_EINVAL, "amis_forward: mc_samples (%d) must be a positive multiple of num_iter (%d)",
                am->mc_samples, am->num_iter);
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose_opt || !pose_cov || !pose_samples || !logweights) return fail(EPROPNP_EINVAL, "amis_forward: NULL pointer");
  {   // default: projection on the matrix cores (amis_forward_mfma.hip); EPROPNP_TUNE="fwd_impl=valu" keeps the VALU sweep
    const char* impl = tune_value("fwd_impl");
    if (!(impl && impl[0] == 'v'))
      return launch_amis_forward_mfma(prob, am, pose_opt, pose_cov, noise, pose_samples, logweights, proposals, st, dn);
  }
  if (prob->num_pts > kMaxResidentPoints)
    return fail(EPROPNP_EINVAL, "amis_forward: num_pts %d exceeds the register-resident limit %d", prob->num_pts,
                kMaxResidentPoints);
  const Problem d = to_device_problem(prob);
  const int S = am->mc_samples, K = am->num_iter, s = S / K;
  const int PL = prob->dof == 6 ? 7 : 4;
  // shape: WP waves split the points (as few as possible: <= 8 points per lane), WS waves split the sample tiles
  int WP = 1;
  while (64 * WP * 8 < d.N && WP < 16) WP *= 2;
  int ppl = 1;
  while (64 * WP * ppl < d.N) ppl *= 2;
  const int ntile = (s + 63) / 64;
  int WS = 1;
  while (WS * 2 <= ntile && WS * 2 * WP <= 8) WS *= 2;
  // few objects: trade points-per-lane for more point-waves to fill the machine (<= 8 waves: no register spills)
  while ((long)d.B * WS * WP < 2048 && ppl > 1 && WS * WP * 2 <= 8) {
    WP *= 2;
    ppl /= 2;
  }
  int ov[3];   // WS, WP, PPL
  if (tune_ints("fwd_shape", ov, 3) && valid_shape_override(ov[1], ov[2], d.N) && ov[0] >= 1 && ov[0] * ov[1] <= 16 &&
      (ov[0] & (ov[0] - 1)) == 0) {
    WS = ov[0]; WP = ov[1]; ppl = ov[2];
  }
  AmisParams k;
  k.split_timeout = 0;
  k.S = S; k.K = K; k.WP = WP; k.eps = am->eps; k.mle_iter = am->acg_mle_iter; k.dispersion = am->acg_dispersion;
  k.seed = am->seed; k.offset = am->offset; k.offset_dev = (const unsigned long long*)am->offset_dev;
  k.advance = (am->advance && am->advance_ticket && am->advance_count > 0) ? (unsigned long long*)am->advance : nullptr;
  k.advance_ticket = (int*)am->advance_ticket; k.advance_count = am->advance_count;
  k.ablate = 0;
  k.dn_offset = dn ? dn->offset : nullptr; k.dn_samples = dn ? dn->samples : nullptr; k.dn_pose_opt = dn ? dn->pose_opt : nullptr;
  { int ab[1]; if (tune_ints("ablate", ab, 1)) k.ablate = ab[0]; }
  // the float4-viewed arrays (ptab rows, wred) come first so that they are 16-B aligned for any S
  const size_t smem = sizeof(float) * (12 * (size_t)s + (size_t)PL * S + 3 * (size_t)S + (size_t)WP * s +
                                       (size_t)K * kPropStride + 256 + (size_t)WS * WP * kWaveRed);
  if (smem > 160 * 1024) return fail(EPROPNP_EINVAL, "amis_forward: mc_samples %d needs %zu B of LDS (> 160 KiB)", S, smem);
  const dim3 grid(padded_object_grid(d.B)), block(64 * WS * WP);
  dispatch_shape(prob->dof, ppl, has_bounds(prob), WS * WP, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    auto kern = amis_forward_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value, decltype(MAXW)::value>;
    allow_dynamic_lds((const void*)kern, smem);
    PNP_LAUNCH(kern, grid, block, smem, st, d, k, pose_opt, pose_cov, noise, pose_samples, logweights, proposals);
    return 0;
  });
  return check_launch("amis_forward_kernel");
}

// grad_w2d[b, :, :] += (sum over parts of grad_delta[b, part]) * d delta[b] / d w2d  for a Huber threshold that came from
// AdaptiveHuberPnPCost on this w2d (epropnp_problem.delta_stats): the follow-up launch for the backward variants whose kernel
// does not know the object's whole grad_delta at its end (object split over workgroups; the all-VALU kernel).
// (grid: B x ceil(2N / kDeltaPathSpan) workgroups -- one workgroup per object took 10 us for 32 x 4096 points, a launch of 32
// workgroups on a 256-CU device)
constexpr int kDeltaPathSpan = 1024;
__global__ __launch_bounds__(256) void delta_path_kernel(Problem p, const float* __restrict__ gdelta, int nparts,
                                                          float* __restrict__ gw2d, int spans) {
  const int b = (int)blockIdx.x / spans, sp = (int)blockIdx.x - b * spans;
  float g = 0.f;
  for (int q = 0; q < nparts; ++q) g += gdelta[(size_t)b * nparts + q];
  const float add = (g * p.delta_stats[(size_t)b * 4 + 1]) * (p.delta_relative / (2.0f * (float)p.N));
  float* row = gw2d + (size_t)b * p.N * 2;
  const int end = min(2 * p.N, (sp + 1) * kDeltaPathSpan);
  for (int i = sp * kDeltaPathSpan + (int)threadIdx.x; i < end; i += (int)blockDim.x) row[i] += add;
}

int launch_delta_path(const epropnp_problem* prob, const float* gdelta, int nparts, float* gw2d, hipStream_t st) {
  if (prob->delta_stats == nullptr) return EPROPNP_OK;
  const Problem d = to_device_problem(prob);
  const int spans = (2 * d.N + kDeltaPathSpan - 1) / kDeltaPathSpan;
  PNP_LAUNCH(delta_path_kernel, dim3((unsigned)(d.B * spans)), dim3(256), 0, st, d, gdelta, nparts, gw2d, spans);
  return check_launch("delta_path_kernel");
}

// Few objects: one object's S x N point-poses keep a single CU busy for ~70 us at 512 x 512 whatever the wave count, so
// the point chunks of an object are dealt to `nsplit` workgroups (each builds the pose table for itself).  The per-point
// gradients are disjoint and bit-identical to the unsplit kernel; grad_delta comes back as (B, nsplit) partials for the
// caller to add in a fixed order (no atomics: results stay reproducible).
int launch_amis_backward_split(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                               int mc_samples, const float* pose_init, const float* grad_cost_init, int nsplit,
                               float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta_parts, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0 || prob->num_pts == 0) return EPROPNP_OK;
  if (mc_samples < 0) return fail(EPROPNP_EINVAL, "amis_backward_split: negative mc_samples");
  if ((mc_samples > 0 && (!pose_samples || !grad_logweights)) || !grad_x3d || !grad_x2d || !grad_w2d || !grad_delta_parts)
    return fail(EPROPNP_EINVAL, "amis_backward_split: NULL pointer");
  if (nsplit < 1 || nsplit > 16 || (long long)nsplit * 64 > (long long)((prob->num_pts + 63) / 64) * 64)
    return fail(EPROPNP_EINVAL, "amis_backward_split: nsplit %d not in [1, min(16, ceil(num_pts / 64))]", nsplit);
  const int rc = launch_amis_backward_mfma(prob, pose_samples, grad_logweights, mc_samples, pose_init, grad_cost_init,
                                           grad_x3d, grad_x2d, grad_w2d, grad_delta_parts, nsplit, st);
  if (rc == 1) return fail(EPROPNP_EINVAL, "amis_backward_split: mc_samples %d does not fit the LDS pose table", mc_samples);
  if (rc != 0) return rc;
  return nsplit > 1 ? launch_delta_path(prob, grad_delta_parts, nsplit, grad_w2d, st) : EPROPNP_OK;      // (1 part: the kernel's epilogue)
}

int launch_amis_backward(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                         int mc_samples, const float* pose_init, const float* grad_cost_init, float* grad_x3d,
                         float* grad_x2d, float* grad_w2d, float* grad_delta, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0 || prob->num_pts == 0) return EPROPNP_OK;
  if (mc_samples < 0) return fail(EPROPNP_EINVAL, "amis_backward: negative mc_samples");
  if ((mc_samples > 0 && (!pose_samples || !grad_logweights)) || !grad_x3d || !grad_x2d || !grad_w2d || !grad_delta)
    return fail(EPROPNP_EINVAL, "amis_backward: NULL pointer");
  {   // projection on the matrix cores (amis_backward_mfma.hip) unless its LDS pose table does not fit;
      // EPROPNP_TUNE="bwd_impl=valu" forces this file's all-VALU kernel
    const char* impl = tune_value("bwd_impl");
    if (!(impl && impl[0] == 'v')) {
      const int rc = launch_amis_backward_mfma(prob, pose_samples, grad_logweights, mc_samples, pose_init, grad_cost_init,
                                               grad_x3d, grad_x2d, grad_w2d, grad_delta, 1, st);
      if (rc <= 0) return rc;     // 1 = shape not supported there (pose table larger than LDS)
    }
  }
  if (prob->num_pts > kMaxResidentPoints)
    return fail(EPROPNP_EINVAL, "amis_backward: num_pts %d exceeds the register-resident limit %d", prob->num_pts,
                kMaxResidentPoints);
  const Problem d = to_device_problem(prob);
  // measured on MI355X (profiles/): fewest waves per object wins (per-pose overhead amortised over 8 points/lane)
  Shape s = choose_shape(d.B, d.N, /*max_ppl=*/8, /*want_waves_total=*/4096);
  int ov[2];
  if (tune_ints("bwd_shape", ov, 2) && valid_shape_override(ov[0], ov[1], d.N)) { s.waves = ov[0]; s.ppl = ov[1]; }
  const dim3 grid(padded_object_grid(d.B)), block(64 * s.waves);
  dispatch_shape(prob->dof, s.ppl, has_bounds(prob), s.waves, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    PNP_LAUNCH((amis_backward_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value, decltype(MAXW)::value>),
               grid, block, 0, st, d, pose_samples, grad_logweights, mc_samples, pose_init, grad_cost_init, grad_x3d,
               grad_x2d, grad_w2d, grad_delta, backward_drop_eps());
    return 0;
  });
  if (int rc = check_launch("amis_backward_kernel")) return rc;
  return launch_delta_path(prob, grad_delta, 1, grad_w2d, st);
}

}  // namespace pnp
