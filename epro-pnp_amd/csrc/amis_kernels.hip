// amis_kernels.hip -- the AMIS Monte-Carlo pose sampler (forward) and its gradient (backward) for gfx950.
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
#include "pnp_host.h"

namespace pnp {

#ifndef PNP_SWEEP_PIPELINE
#define PNP_SWEEP_PIPELINE 0
#endif

constexpr int kPropStride = 40;   // floats per fitted proposal (layout below)
// proposal record:  [0..2] t-mode | [3..8] L_t (lower, row-major packed) | [9..14] L_t^-1 | [15] Student-t log-norm
//   6-DoF: [16..25] L_r (lower 4x4 packed) | [26..35] L_r^-1 | [36] sum log diag L_r
//   4-DoF: [16] yaw mode | [17] kappa | [18] log I0(kappa)
constexpr int kVmTries = 16;
constexpr int kRedStride = 68;                      // 64 lanes + 4 floats of padding per parked sample
constexpr int kWaveRed = 16 * kRedStride + 64;     // per-wave LDS scratch of the transposed cost reduction

struct AmisParams {
  int S, K;            // total samples, iterations
  int WP;              // waves that split the points (W = WS * WP)
  float eps;
  int mle_iter;
  float dispersion;
  unsigned long long seed, offset;
  int ablate;          // tuning builds only (-DPNP_TUNING): bit0 skip sweep, bit1 skip proposal refit, bit2 skip densities
};

// The fp64 proposal fits run on one lane a handful of times per object; keeping them out of line stops their
// ~100 live fp64 registers from inflating the allocation of the VALU-bound sweep loops (occupancy).
#ifndef PNP_FIT_FN
#define PNP_FIT_FN __device__ __forceinline__
#endif

__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// ---- proposal fitting helpers (run by thread 0 only; fp64 so that the ill-conditioned 4x4 inversions of the
// ---- reference's fp32 LAPACK path are at least not made worse) ----------------------------------------------

// pack Cholesky factor / its inverse / log-normaliser of a 3x3 translation covariance into rec[3..15]
PNP_FIT_FN void fit_translation(double (&C)[3][3], const float* fallback_diag, float* rec) {
  double invd[3];
  const bool ok = cholesky<3, double>(C, invd);
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) C[i][j] = (i == j) ? (double)fallback_diag[i] : 0.0;
      invd[i] = 1.0 / (double)fallback_diag[i];
    }
  }
  double Li[3][3];
  tri_inverse<3, double>(C, invd, Li);
  float sl = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sl += logf((float)C[i][i]);
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      rec[3 + tri(i, j)] = (float)C[i][j];
      rec[9 + tri(i, j)] = (float)Li[i][j];
    }
  }
  rec[15] = student_t3_log_norm(sl);
}

// rot_cov (4x4 SPD, trace ~ 1) -> + det^(1/4) * dispersion * I -> Cholesky -> rec[16..36]   (epropnp.py:301-302,341-342)
PNP_FIT_FN void fit_rotation_acg(double (&Rc)[4][4], float dispersion, float* rec) {
  double Lc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) Lc[i][j] = Rc[i][j];
  double invd[4];
  bool ok = cholesky<4, double>(Lc, invd);
  // det^(1/4) = sqrt(prod of the Cholesky pivots).  reference: torch.det on a possibly indefinite matrix; in the
  // non-SPD case any value leads to the Cholesky fallback below, so the SPD determinant is all that matters
  const float pivots = (float)(Lc[0][0] * Lc[1][1] * Lc[2][2] * Lc[3][3]);
  const double add = ok ? (double)(sqrtf(pivots) * dispersion) : 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) Rc[i][i] += add;
  ok = cholesky<4, double>(Rc, invd) && ok;
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) Rc[i][j] = (i == j) ? 1.0 : 0.0;
      invd[i] = 1.0;
    }
  }
  double Li[4][4];
  tri_inverse<4, double>(Rc, invd, Li);
  float sl = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sl += logf((float)Rc[i][i]);
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      rec[16 + tri(i, j)] = (float)Rc[i][j];
      rec[26 + tri(i, j)] = (float)Li[i][j];
    }
  }
  rec[36] = sl;
}

// proposal #0 from the Laplace approximation at the LM solution
template <int DOF>
PNP_FIT_FN void initial_fit(const float* pose_opt, const float* cov, float eps, float dispersion, float* rec) {
  rec[0] = pose_opt[0]; rec[1] = pose_opt[1]; rec[2] = pose_opt[2];
  double Ct[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Ct[i][j] = (double)cov[i * DOF + j];
  if (DOF == 4) {
    const float dflt[3] = {1.0f, 1.0f, 4.0f};
    fit_translation(Ct, dflt, rec);
    rec[16] = pose_opt[3];
    const float kappa = 0.33f / fmaxf(cov[3 * 4 + 3], eps);
    rec[17] = kappa;
    rec[18] = log_i0(kappa);
  } else {
    const float dflt[3] = {1.0f, 1.0f, 1.0f};
    fit_translation(Ct, dflt, rec);
    double Cr[3][3], Ci[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Cr[i][j] = (double)cov[(3 + i) * 6 + 3 + j];
    double invd3[3], invd4[4];
    spd_inverse<3, double>(Cr, invd3, Ci);
    const double w = pose_opt[3], qi = pose_opt[4], qj = pose_opt[5], qk = pose_opt[6];
    const double T[4][3] = {{qi, qj, qk}, {-w, -qk, qj}, {qk, -w, -qi}, {-qj, qi, -w}};   // camera.py:158-165
    double TC[4][3], A[4][4], Ai[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) TC[i][j] = T[i][0] * Ci[0][j] + T[i][1] * Ci[1][j] + T[i][2] * Ci[2][j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        A[i][j] = TC[i][0] * T[j][0] + TC[i][1] * T[j][1] + TC[i][2] * T[j][2] + ((i == j) ? 1.0 : 0.0);
    spd_inverse<4, double>(A, invd4, Ai);
    const double itr = 1.0 / (Ai[0][0] + Ai[1][1] + Ai[2][2] + Ai[3][3]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Ai[i][j] *= itr;
    fit_rotation_acg(Ai, dispersion, rec);
  }
}

// log q_j(sample) for proposal record `rec`; sample components passed in registers
template <int DOF>
PNP_FN float proposal_logprob(const float* rec, const float* smp /*PL*/) {
  const float d0 = smp[0] - rec[0], d1 = smp[1] - rec[1], d2 = smp[2] - rec[2];
  const float y0 = rec[9] * d0;
  const float y1 = rec[10] * d0 + rec[11] * d1;
  const float y2 = rec[12] * d0 + rec[13] * d1 + rec[14] * d2;
  float lp = student_t3_logprob(y0 * y0 + y1 * y1 + y2 * y2, rec[15]);
  if (DOF == 6) {
    const float a = smp[3], b = smp[4], c = smp[5], d = smp[6];
    const float r0 = rec[26] * a;
    const float r1 = rec[27] * a + rec[28] * b;
    const float r2 = rec[29] * a + rec[30] * b + rec[31] * c;
    const float r3 = rec[32] * a + rec[33] * b + rec[34] * c + rec[35] * d;
    lp += acg4_logprob(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3, rec[36]);
  } else {
    lp += vm_mix_logprob(smp[3], rec[16], rec[17], rec[18]);
  }
  return lp;
}

// Best & Fisher (1979) von Mises draw with at most kVmTries attempts; u = 3 uniforms per attempt.
// Same bounded procedure as oracle/epropnp_oracle.py:vm_sample_bounded.
PNP_FN float vm_sample_bounded(float loc, float kappa, const float* u /* kVmTries*3 */) {
  const double k = fmax((double)kappa, 1e-12);
  const double tau = 1.0 + sqrt(1.0 + 4.0 * k * k);
  const double rho = (tau - sqrt(2.0 * tau)) / (2.0 * k);
  const double r = (k < 1e-5) ? (1.0 / k + k) : (1.0 + rho * rho) / (2.0 * rho);
  double x = 0.0;
  bool done = false;
  for (int a = 0; a < kVmTries; ++a) {
    const double u1 = u[a * 3], u2 = u[a * 3 + 1], u3 = u[a * 3 + 2];
    const double zc = cos(3.141592653589793 * u1);
    const double f = (1.0 + r * zc) / (r + zc);
    const double c = k * (r - f);
    const bool acc = ((c * (2.0 - c) - u2) > 0.0) || ((log(c / fmax(u2, 1e-300)) + 1.0 - c) >= 0.0);
    if (!done && (acc || a == kVmTries - 1)) {
      x = ((u3 - 0.5 >= 0.0) ? 1.0 : -1.0) * acos(fmin(fmax(f, -1.0), 1.0));
    }
    done = done || acc;
  }
  double y = x + 3.141592653589793 + (double)loc;
  y = y - 6.283185307179586 * floor(y / 6.283185307179586);
  return (float)(y - 3.141592653589793);
}

// ================================================================================================================
// forward
// ================================================================================================================
template <int DOF, int PPL, bool BOUNDS, int MAXW>
__global__ __launch_bounds__(MAXW * 64, (MAXW == 4 ? (PPL >= 8 ? 2 : 3) : 1)) void amis_forward_kernel(Problem p, AmisParams a,
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

  if (tid == 0) initial_fit<DOF>(pose_opt + (size_t)b * PL, pose_cov + (size_t)b * DOF * DOF, a.eps, a.dispersion, prop);
  __syncthreads();

  for (int it = 0; it < K; ++it) {
    const float* rec = prop + it * kPropStride;
    // ---------------- 1. draw s samples from proposal `it` (lane = sample) ----------------
    for (int n = tid; n < s; n += T) {
      const int m = it * s + n;
      float z[3], chi2, g[4], uvm[3 * kVmTries];
      if (noise != nullptr) {
        const float* nz = noise + (((size_t)b * K + it) * s + n) * NZ;
        z[0] = nz[0]; z[1] = nz[1]; z[2] = nz[2]; chi2 = nz[3];
        if (DOF == 6) {
          g[0] = nz[4]; g[1] = nz[5]; g[2] = nz[6]; g[3] = nz[7];
        } else {
#pragma unroll
          for (int i = 0; i < 3 * kVmTries; ++i) uvm[i] = nz[4 + i];
        }
      } else {
        const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
        const uint32_t c2 = (uint32_t)a.offset, c3base = (uint32_t)(a.offset >> 32) * 64u;
        float nrm[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const Philox4 r = philox4x32_10((uint32_t)b, (uint32_t)m, c2, c3base + q, k0, k1);
          box_muller(r.v[0], r.v[1], nrm[q * 4], nrm[q * 4 + 1]);
          box_muller(r.v[2], r.v[3], nrm[q * 4 + 2], nrm[q * 4 + 3]);
        }
        z[0] = nrm[0]; z[1] = nrm[1]; z[2] = nrm[2];
        chi2 = nrm[3] * nrm[3] + nrm[4] * nrm[4] + nrm[5] * nrm[5];   // Chi2(3)
        if (DOF == 6) {
          g[0] = nrm[6]; g[1] = nrm[7]; g[2] = nrm[8]; g[3] = nrm[9];
        } else {
#pragma unroll
          for (int q = 0; q < (3 * kVmTries) / 4; ++q) {
            const Philox4 r = philox4x32_10((uint32_t)b, (uint32_t)m, c2, c3base + 8 + q, k0, k1);
#pragma unroll
            for (int e = 0; e < 4; ++e) uvm[q * 4 + e] = (float)(r.v[e] >> 8) * (1.0f / 16777216.0f);
          }
        }
      }
      float ps[PL];
      // translation: mode + L_t (z * rsqrt(chi2 / 3))
      const float sc = 1.0f / sqrtf(chi2 / 3.0f);
      const float y0 = z[0] * sc, y1 = z[1] * sc, y2 = z[2] * sc;
      ps[0] = rec[0] + rec[3] * y0;
      ps[1] = rec[1] + (rec[4] * y0 + rec[5] * y1);
      ps[2] = rec[2] + (rec[6] * y0 + rec[7] * y1 + rec[8] * y2);
      if (DOF == 6) {   // ACG: L_r g / |L_r g|   (distributions.py:42-52)
        const float v0 = rec[16] * g[0];
        const float v1 = rec[17] * g[0] + rec[18] * g[1];
        const float v2 = rec[19] * g[0] + rec[20] * g[1] + rec[21] * g[2];
        const float v3 = rec[22] * g[0] + rec[23] * g[1] + rec[24] * g[2] + rec[25] * g[3];
        const float nr = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3);
        if (nr < 1e-6f) {
          ps[3] = 1.f; ps[4] = 0.f; ps[5] = 0.f; ps[6] = 0.f;
        } else {
          ps[3] = v0 / nr; ps[4] = v1 / nr; ps[5] = v2 / nr; ps[6] = v3 / nr;
        }
      } else {          // first round(0.25 s) samples uniform, the rest von Mises  (distributions.py:65-71)
        const int n_u = (int)rintf(0.25f * (float)s);
        if (n < n_u) ps[3] = (uvm[0] * 2.0f - 1.0f) * 3.14159265358979f;
        else ps[3] = vm_sample_bounded(rec[16], rec[17], uvm);
      }
#pragma unroll
      for (int i = 0; i < PL; ++i) {
        smp[i * S + m] = ps[i];
        pose_samples[((size_t)m * p.B + b) * PL + i] = ps[i];
      }
      {   // project_b operands of this sample -> LDS row (read back as broadcast by every lane of the sweep)
        float R[9], KR[9], Kt[3];
        pose_to_rot<DOF>(ps, R);
        compose_kr_kt(Kc, R, ps, KR, Kt);
        float4* row = reinterpret_cast<float4*>(ptab + 12 * n);
        row[0] = make_float4(KR[0], KR[1], KR[2], KR[3]);
        row[1] = make_float4(KR[4], KR[5], KR[6], KR[7]);
        row[2] = make_float4(KR[8], Kt[0], Kt[1], Kt[2]);
      }
    }
    __syncthreads();

    // ---------------- 2. cost sweep: s poses x this wave's points (lane = point) ----------------
    const int ntile = (s + 63) >> 6;
#ifdef PNP_TUNING
    if (a.ablate & 1) {
      for (int n = tid; n < s; n += T) cpart[n] = 1.0f;
    } else
#endif
    for (int t = ws; t < ntile; t += WS) {
      const int base = t * 64;
      const int cnt = min(64, s - base);
      float mine = 0.f;
#if PNP_SWEEP_PIPELINE
      // two samples per trip (their DPP reduction chains interleave); the next trip's first pose row is fetched
      // from LDS before the current pair is evaluated, so the ds_read latency hides behind ~400 VALU instructions
      const float4* row = reinterpret_cast<const float4*>(ptab + 12 * base);   // uniform address: LDS broadcast
      float4 n0 = row[0], n1 = row[1], n2 = row[2];
      for (int j = 0; j < cnt; j += 2) {
        const float4 a0 = n0, a1 = n1, a2 = n2;
        const float4* rb = reinterpret_cast<const float4*>(ptab + 12 * (base + min(j + 1, cnt - 1)));
        const float4 b0 = rb[0], b1 = rb[1], b2 = rb[2];
        const float4* rn = reinterpret_cast<const float4*>(ptab + 12 * (base + min(j + 2, cnt - 1)));
        n0 = rn[0]; n1 = rn[1]; n2 = rn[2];
        const float krA[9] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x};
        const float ktA[3] = {a2.y, a2.z, a2.w};
        const float krB[9] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x};
        const float ktB[3] = {b2.y, b2.z, b2.w};
        float cA = 0.f, cB = 0.f;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          cA += sweep_cost<BOUNDS>(pts[k], krA, ktA, zmin_v, delta_v, bd);
          cB += sweep_cost<BOUNDS>(pts[k], krB, ktB, zmin_v, delta_v, bd);
        }
        cA = wave_sum(cA);
        cB = wave_sum(cB);
        mine = (lane == j) ? cA : mine;
        mine = (lane == j + 1) ? cB : mine;     // j + 1 == cnt (odd tail) only ever matches a lane >= cnt: unused
      }
#else
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
          const float kr[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x};
          const float kt[3] = {r2.y, r2.z, r2.w};
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
#endif
      if (base + lane < s) cpart[wp * s + base + lane] = mine;
    }
    __syncthreads();

    // ---------------- 3+4. proposal densities, mixture, log-weights (lane = sample) ----------------
    const int M = (it + 1) * s;
    const float log_n = logf((float)(it + 1));
    for (int m = tid; m < M; m += T) {
      float ps[PL];
#pragma unroll
      for (int i = 0; i < PL; ++i) ps[i] = smp[i * S + m];
      float mix;
      if (m >= it * s) {   // new sample: every proposal so far
        float c = cpart[m - it * s];
        for (int q = 1; q < WP; ++q) c += cpart[q * s + (m - it * s)];
        cst[m] = c;
#ifdef PNP_TUNING
        if (a.ablate & 4) mix = 0.f; else {
#endif
        mix = proposal_logprob<DOF>(prop, ps);
        for (int j = 1; j <= it; ++j) mix = log_add_exp(mix, proposal_logprob<DOF>(prop + j * kPropStride, ps));
#ifdef PNP_TUNING
        }
#endif
      } else {             // old sample: add the new proposal's density
#ifdef PNP_TUNING
        if (a.ablate & 4) mix = 0.f; else
#endif
        mix = log_add_exp(mixl[m], proposal_logprob<DOF>(rec, ps));
      }
      mixl[m] = mix;
      lgw[m] = -cst[m] - (mix - log_n);
    }
    __syncthreads();
    if (it == K - 1) break;

    // ---------------- 5. fit proposal it+1 to the weighted samples (epropnp.py:238-260 / :317-342) ---------
    float* nrec = prop + (it + 1) * kPropStride;
#ifdef PNP_TUNING
    if (a.ablate & 2) {
      for (int i = tid; i < kPropStride; i += T) nrec[i] = rec[i];
      __syncthreads();
      continue;
    }
#endif
    float mx = -INFINITY;
    for (int m = tid; m < M; m += T) mx = fmaxf(mx, lgw[m]);
    mx = block_max(mx, red);
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = tid; m < M; m += T) {
      const float e = expf(lgw[m] - mx);
      s4[0] += e;
      s4[1] += e * smp[0 * S + m];
      s4[2] += e * smp[1 * S + m];
      s4[3] += e * smp[2 * S + m];
    }
    block_sum<4>(s4, red);
    const float invZ = 1.0f / s4[0];
    const float mu0 = s4[1] * invZ, mu1 = s4[2] * invZ, mu2 = s4[3] * invZ;
    if (DOF == 6) {
      float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int m = tid; m < M; m += T) {
        const float w = expf(lgw[m] - mx) * invZ;
        const float d0 = smp[m] - mu0, d1 = smp[S + m] - mu1, d2 = smp[2 * S + m] - mu2;
        c6[0] += w * d0 * d0; c6[1] += w * d1 * d0; c6[2] += w * d1 * d1;
        c6[3] += w * d2 * d0; c6[4] += w * d2 * d1; c6[5] += w * d2 * d2;
      }
      block_sum<6>(c6, red);
      // ACG maximum-likelihood fixed point, Sigma_0 = I
      float Si[10] = {1.f, 0.f, 1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};   // packed lower of Sigma^-1
      float acc[11];
      for (int r = 0; r < a.mle_iter; ++r) {
#pragma unroll
        for (int i = 0; i < 11; ++i) acc[i] = 0.f;
        for (int m = tid; m < M; m += T) {
          const float w = expf(lgw[m] - mx) * invZ;
          const float q0 = smp[3 * S + m], q1 = smp[4 * S + m], q2 = smp[5 * S + m], q3 = smp[6 * S + m];
          const float Mq = Si[0] * q0 * q0 + Si[2] * q1 * q1 + Si[5] * q2 * q2 + Si[9] * q3 * q3 +
                           2.f * (Si[1] * q1 * q0 + Si[3] * q2 * q0 + Si[4] * q2 * q1 + Si[6] * q3 * q0 + Si[7] * q3 * q1 +
                                  Si[8] * q3 * q2);
          const float iw = w / fmaxf(Mq, a.eps);
          acc[10] += iw;
          acc[0] += iw * q0 * q0;
          acc[1] += iw * q1 * q0; acc[2] += iw * q1 * q1;
          acc[3] += iw * q2 * q0; acc[4] += iw * q2 * q1; acc[5] += iw * q2 * q2;
          acc[6] += iw * q3 * q0; acc[7] += iw * q3 * q1; acc[8] += iw * q3 * q2; acc[9] += iw * q3 * q3;
        }
        block_sum<11>(acc, red);
        if (r + 1 < a.mle_iter) {   // need Sigma^-1 for the next fixed-point step
          if (tid == 0) {
            double Sg[4][4], Sgi[4][4], invd[4];
            const double inorm = 1.0 / (double)acc[10];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j <= i; ++j) {
                const double v = (double)acc[tri(i, j)] * inorm + ((i == j) ? (double)a.eps : 0.0);
                Sg[i][j] = v;
                Sg[j][i] = v;
              }
            spd_inverse<4, double>(Sg, invd, Sgi);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j <= i; ++j) red[tri(i, j)] = (float)Sgi[i][j];
          }
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 10; ++i) Si[i] = red[i];
          __syncthreads();
        }
      }
      if (tid == 0) {
        nrec[0] = mu0; nrec[1] = mu1; nrec[2] = mu2;
        double Ct[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            Ct[i][j] = (double)c6[tri(i, j)];
            Ct[j][i] = (double)c6[tri(i, j)];
          }
        const float dflt[3] = {1.f, 1.f, 1.f};
        fit_translation(Ct, dflt, nrec);
        double Sg[4][4];
        const double inorm = (a.mle_iter > 0) ? 1.0 / (double)acc[10] : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            double v = (a.mle_iter > 0) ? (double)acc[tri(i, j)] * inorm + ((i == j) ? (double)a.eps : 0.0)
                                        : ((i == j) ? 1.0 : 0.0);
            Sg[i][j] = v;
            Sg[j][i] = v;
          }
        fit_rotation_acg(Sg, a.dispersion, nrec);
      }
    } else {
      float c8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int m = tid; m < M; m += T) {
        const float w = expf(lgw[m] - mx) * invZ;
        const float d0 = smp[m] - mu0, d1 = smp[S + m] - mu1, d2 = smp[2 * S + m] - mu2;
        c8[0] += w * d0 * d0; c8[1] += w * d1 * d0; c8[2] += w * d1 * d1;
        c8[3] += w * d2 * d0; c8[4] += w * d2 * d1; c8[5] += w * d2 * d2;
        const float yaw = smp[3 * S + m];
        c8[6] += w * sinf(yaw);
        c8[7] += w * cosf(yaw);
      }
      block_sum<8>(c8, red);
      if (tid == 0) {
        nrec[0] = mu0; nrec[1] = mu1; nrec[2] = mu2;
        double Ct[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            Ct[i][j] = (double)c8[tri(i, j)];
            Ct[j][i] = (double)c8[tri(i, j)];
          }
        const float dflt[3] = {1.f, 1.f, 4.f};
        fit_translation(Ct, dflt, nrec);
        nrec[16] = atan2f(c8[6], c8[7]);
        const float r_sq = c8[6] * c8[6] + c8[7] * c8[7];
        const float kappa = 0.33f * fmaxf(sqrtf(r_sq), a.eps) * (2.f - r_sq) / fmaxf(1.f - r_sq, a.eps);
        nrec[17] = kappa;
        nrec[18] = log_i0(kappa);
      }
    }
    __syncthreads();
  }

  // ---------------- outputs ----------------
  for (int m = tid; m < S; m += T) logweights[(size_t)m * p.B + b] = lgw[m];
  if (proposals != nullptr)
    for (int i = tid; i < K * kPropStride; i += T) proposals[(size_t)b * K * kPropStride + i] = prop[i];
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
                                                                    float* __restrict__ gw2d, float* __restrict__ gdelta) {
  constexpr int PL = PoseLen<DOF>::value;
  // pose tiles: 64 poses x {K R (9) | K t (3) | weight | pad} as 4 float4 rows, double buffered
  __shared__ __attribute__((aligned(16))) float tab[2][64][16];
  __shared__ float red[16];
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

  // lanes 0..63 of the workgroup fetch one pose each of tile t (global loads issued early, consumed late)
  float nps[PL], naw = 0.f;
  auto fetch = [&](int t) {
    if (tid < 64) {
      const int m = min(t * 64 + tid, P - 1);
      const float* src = (m < S) ? pose_samples + ((size_t)m * p.B + b) * PL : pose_init + (size_t)b * PL;
#pragma unroll
      for (int i = 0; i < PL; ++i) nps[i] = src[i];
      naw = (m < S) ? -g_logw[(size_t)m * p.B + b] : g_init[b];     // logw = -cost - const
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
#ifndef EPROPNP_EMU
      if (__builtin_amdgcn_readfirstlane(__float_as_int(a)) == 0) continue;   // exact zero weight (wave-uniform)
#else
      if (a == 0.f) continue;
#endif
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
          px = fminf(fmaxf(px, bd.lbx), bd.ubx);
          py = fminf(fmaxf(py, bd.lby), bd.uby);
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
                        float* proposals, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (!am) return fail(EPROPNP_EINVAL, "amis_forward: params NULL");
  if (am->num_iter <= 0 || am->mc_samples <= 0 || am->mc_samples % am->num_iter != 0)
    return fail(EPROPNP_EINVAL, "amis_forward: mc_samples (%d) must be a positive multiple of num_iter (%d)",
                am->mc_samples, am->num_iter);
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose_opt || !pose_cov || !pose_samples || !logweights) return fail(EPROPNP_EINVAL, "amis_forward: NULL pointer");
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
  if (env_ints("EPROPNP_FWD_SHAPE", ov, 3) && 64 * ov[1] * ov[2] >= d.N && ov[0] * ov[1] <= 16) {
    WS = ov[0]; WP = ov[1]; ppl = ov[2];
  }
  AmisParams k;
  k.S = S; k.K = K; k.WP = WP; k.eps = am->eps; k.mle_iter = am->acg_mle_iter; k.dispersion = am->acg_dispersion;
  k.seed = am->seed; k.offset = am->offset;
  k.ablate = 0;
  { int ab[1]; if (env_ints("EPROPNP_ABLATE", ab, 1)) k.ablate = ab[0]; }
  // the float4-viewed arrays (ptab rows, wred) come first so that they are 16-B aligned for any S
  const size_t smem = sizeof(float) * (12 * (size_t)s + (size_t)PL * S + 3 * (size_t)S + (size_t)WP * s +
                                       (size_t)K * kPropStride + 256 + (size_t)WS * WP * kWaveRed);
  if (smem > 160 * 1024) return fail(EPROPNP_EINVAL, "amis_forward: mc_samples %d needs %zu B of LDS (> 160 KiB)", S, smem);
  const dim3 grid(padded_object_grid(d.B)), block(64 * WS * WP);
  dispatch_shape(prob->dof, ppl, has_bounds(prob), WS * WP, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    auto kern = amis_forward_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value, decltype(MAXW)::value>;
#ifndef EPROPNP_EMU
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
    PNP_LAUNCH(kern, grid, block, smem, st, d, k, pose_opt, pose_cov, noise, pose_samples, logweights, proposals);
    return 0;
  });
  return check_launch("amis_forward_kernel");
}

int launch_amis_backward(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                         int mc_samples, const float* pose_init, const float* grad_cost_init, float* grad_x3d,
                         float* grad_x2d, float* grad_w2d, float* grad_delta, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0 || prob->num_pts == 0) return EPROPNP_OK;
  if (mc_samples < 0) return fail(EPROPNP_EINVAL, "amis_backward: negative mc_samples");
  if ((mc_samples > 0 && (!pose_samples || !grad_logweights)) || !grad_x3d || !grad_x2d || !grad_w2d || !grad_delta)
    return fail(EPROPNP_EINVAL, "amis_backward: NULL pointer");
  if (prob->num_pts > kMaxResidentPoints)
    return fail(EPROPNP_EINVAL, "amis_backward: num_pts %d exceeds the register-resident limit %d", prob->num_pts,
                kMaxResidentPoints);
  const Problem d = to_device_problem(prob);
  // measured on MI355X (profiles/): fewest waves per object wins (per-pose overhead amortised over 8 points/lane)
  Shape s = choose_shape(d.B, d.N, /*max_ppl=*/8, /*want_waves_total=*/4096);
  int ov[2];
  if (env_ints("EPROPNP_BWD_SHAPE", ov, 2) && 64 * ov[0] * ov[1] >= d.N) { s.waves = ov[0]; s.ppl = ov[1]; }
  const dim3 grid(padded_object_grid(d.B)), block(64 * s.waves);
  dispatch_shape(prob->dof, s.ppl, has_bounds(prob), s.waves, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    PNP_LAUNCH((amis_backward_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value, decltype(MAXW)::value>),
               grid, block, 0, st, d, pose_samples, grad_logweights, mc_samples, pose_init, grad_cost_init, grad_x3d,
               grad_x2d, grad_w2d, grad_delta);
    return 0;
  });
  return check_launch("amis_backward_kernel");
}

}  // namespace pnp
