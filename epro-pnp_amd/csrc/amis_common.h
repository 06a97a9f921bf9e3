// amis_common.h -- pieces of the AMIS sampler shared by the two forward kernels (VALU sweep / MFMA sweep):
// proposal fitting, proposal densities, sampling, weight algebra.  See amis_kernels.hip for the reference map.
#pragma once
#include "pnp_host.h"
#include "tuning.h"

namespace pnp {

typedef PNP_FIT_T fit_t;      // precision of the single-lane proposal fits (tuning.h)

// min(x, 1) for x >= 0 through the clamp output modifier of a multiply by an opaque 1.0 (v_mul_f32 ... clamp, a full-rate
// instruction; v_min_f32 issues at half that rate on gfx950, tools/ubench)
__device__ __forceinline__ float sat_mul(float x, float one_v) {
  return __builtin_amdgcn_fmed3f(x * one_v, 0.0f, 1.0f);
}

// clamp(a * b + c, 0, 1) as one v_fma_f32 ... clamp
__device__ __forceinline__ float sat_fma(float a, float b, float c) {
  return __builtin_amdgcn_fmed3f(fmaf(a, b, c), 0.0f, 1.0f);
}

// max(x, lo) for lo >= 0 as ONE v_max_i32 on the bit patterns: non-negative floats order like their bits and every
// negative x (sign bit set) is a negative integer, so it yields lo (check_problem rejects z_min < 0).  fmaxf on an MFMA
// result is expanded to a canonicalising v_max(x, x) plus the max (IEEE mode).  It has to be an instruction the compiler
// can see: the hazard recogniser pads MFMA -> VALU reads with s_nop, which it cannot do around inline asm -- a
// hand-written v_max_f32 here happened to work behind v_mfma_f32_16x16x4_f32 and read stale registers behind the
// shorter v_mfma_f32_16x16x32_bf16 (profiles/r02_tune_fwd_bf16_split.txt).
__device__ __forceinline__ float clamp_below(float x, float lo) {
  int xi, li;
  memcpy(&xi, &x, 4);
  memcpy(&li, &lo, 4);
  const int mi = max(xi, li);
  float r;
  memcpy(&r, &mi, 4);
  return r;
}

// Residuals in units of the object's Huber threshold: the weights are pre-multiplied by 1 / delta.  The factor is capped at
// 1e12 so that a zero or denormal threshold (degenerate input: all image points equal, all weights zero) neither divides
// by zero nor overflows the squared norm for residuals below 1e7; min(rho, delta) is then min(rho, 1e-12), i.e. a cost
// below 1e-12 rho where the reference has exactly 0.  The threshold is capped at 1e12 on the large side as well: a huge or
// infinite delta (the Huber kernel switched off) would make 1 / delta zero and delta^2 infinite, 0 * inf = NaN costs;
// below rho = 1e12 pixels min(rho, delta) never binds, so the capped threshold computes the same pure quadratic.
struct HuberScale { float inv_delta, delta, delta_sq; };
__device__ __forceinline__ HuberScale huber_scale(float delta) {
  HuberScale h;
  delta = fminf(delta, 1e12f);
  h.inv_delta = fminf(1.0f / delta, 1e12f);
  h.delta = (h.inv_delta < 1e12f) ? delta : 1e-12f;
  h.delta_sq = h.delta * h.delta;
  return h;
}

constexpr int kPropStride = 40;   // floats per fitted proposal (layout below)
// proposal record:  [0..2] t-mode | [3..8] L_t (lower, row-major packed) | [9..14] L_t^-1 | [15] Student-t log-norm
//   6-DoF: [16..25] L_r (lower 4x4 packed) | [26..35] L_r^-1 | [36] sum log diag L_r
//   4-DoF: [16] yaw mode | [17] kappa | [18] log I0(kappa)
//   [37] 1 if the translation covariance fell back to its default factor, [38] 1 if the ACG shape matrix did
//   (cholesky_wrapper, epropnp.py:16-33) -- reported through the status word as EPROPNP_ST_CHOL_FALLBACK
constexpr int kVmTries = PNP_VM_TRIES;             // (tuning.h; the injected-noise layout assumes 16)
constexpr int kRedStride = 68;                      // 64 lanes + 4 floats of padding per parked sample
constexpr int kWaveRed = 16 * kRedStride + 64;     // per-wave LDS scratch of the transposed cost reduction

struct AmisParams {
  int S, K;            // total samples, iterations
  int WP;              // waves that split the points (W = WS * WP)
  float eps;
  int mle_iter;
  float dispersion;
  unsigned long long seed, offset;
  const unsigned long long* offset_dev;   // optional device-side addend to `offset` (graph replay)
  unsigned long long* advance;            // optional: counters the last workgroup to retire increments (advance_counters)
  int* advance_ticket;
  int advance_count;
  int ablate;          // tuning builds only (-DPNP_TUNING): bit0 skip sweep, bit1 skip proposal refit, bit2 skip densities
  unsigned split_timeout;   // split over workgroups: shader cycles a part waits for a sibling's partial costs (wave_ops.h)
  // optional, pnp_normalize'd problems of the one-call forward (mc_forward.hip): pnp_denormalize of the outputs by THIS launch --
  // the samples are stored a second time with translation - R offset (the arithmetic of shift_poses_kernel, sign -1), and
  // pose_opt likewise, instead of a launch of their own behind this one
  const float* dn_offset;   // (B,3) object centres, or nullptr
  float* dn_samples;        // (S,B,PL)
  float* dn_pose_opt;       // (B,PL)
};

// pnp_denormalize of pose_opt (AmisParams.dn_*): one thread of the object's (first) workgroup
template <int DOF>
PNP_FN void denormalise_pose_opt(const AmisParams& a, const float* __restrict__ pose_opt, int b) {
  constexpr int PL = PoseLen<DOF>::value;
  if (a.dn_offset == nullptr || a.dn_pose_opt == nullptr) return;
  float ps[PL], R[9];
#pragma unroll
  for (int i = 0; i < PL; ++i) ps[i] = pose_opt[(size_t)b * PL + i];
  pose_to_rot<DOF>(ps, R);
  shift_translation(ps, R, a.dn_offset[(size_t)b * 3], a.dn_offset[(size_t)b * 3 + 1], a.dn_offset[(size_t)b * 3 + 2], -1.0f);
#pragma unroll
  for (int i = 0; i < PL; ++i) a.dn_pose_opt[(size_t)b * PL + i] = ps[i];
}

// AmisParams.advance: called once by every workgroup that takes part in the launch (`expected` of them), at its very end.  The last
// one to arrive increments the caller's counters and returns the ticket to zero; every workgroup read *offset_dev when it started,
// i.e. before it took its ticket, and the next launch on the stream sees the new values (kernel boundary).  No fences: nothing a
// workgroup wrote is read by another one here, the ticket is a device-scope atomic, and an agent-scope fence per workgroup --
// a write-back of the XCD's L2 -- cost the Det forward 13 us of 43 when it was tried (profiles/r05_launch_fusions.txt).
__device__ __forceinline__ void advance_counters(const AmisParams& a, int expected) {
  if (a.advance == nullptr) return;
  if (threadIdx.x == 0) {
    if (atomicAdd(a.advance_ticket, 1) == expected - 1) {
      for (int i = 0; i < a.advance_count; ++i) atomicAdd(&a.advance[i], 1ull);
      atomicExch(a.advance_ticket, 0);
    }
  }
}

// (PNP_FIT_FN, tuning.h: the fp64 proposal fits run on one lane a handful of times per object)

__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// ---- proposal fitting helpers (run by thread 0 only; fp64 so that the ill-conditioned 4x4 inversions of the
// ---- reference's fp32 LAPACK path are at least not made worse) ----------------------------------------------

// pack Cholesky factor / its inverse / log-normaliser of a 3x3 translation covariance into rec[3..15]
PNP_FIT_FN void fit_translation(fit_t (&C)[3][3], const float* fallback_diag, float* rec) {
  fit_t invd[3];
  const bool ok = cholesky<3, fit_t>(C, invd);
  rec[37] = ok ? 0.f : 1.f;
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) C[i][j] = (i == j) ? (fit_t)fallback_diag[i] : fit_t(0);
      invd[i] = fit_t(1) / (fit_t)fallback_diag[i];
    }
  }
  fit_t Li[3][3];
  tri_inverse<3, fit_t>(C, invd, Li);
  float sl = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sl += fast_log((float)C[i][i]);
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      rec[3 + tri(i, j)] = (float)C[i][j];
      rec[9 + tri(i, j)] = (float)Li[i][j];
    }
  }
  rec[15] = student_t3_log_norm(sl);
}

// ---- the refit's reductions: NV per-lane partial sums -> NV totals, ONE wave, through LDS -------------------------------
// A wave_sum per value costs 11 VALU instructions (4 DPP adds, 4 v_readlane, 3 adds: DPP and readlane issue at half rate),
// 231 for the 21 moments of the 6-DoF refit -- on the one wave the other three of the workgroup are waiting for.  Transposed
// instead: every lane parks its NV partials (row v, column = lane: NV ds_write), lane 4 v' + j then adds columns
// [16 j, 16 j + 16) of row v = base + v' (4 ds_read_b128, 15 adds), two quad-DPP adds join the four quarters, and the NV
// totals come back as broadcast reads: ~35 VALU instructions for 21 values.  Fixed order: bit-reproducible.
constexpr int kRefitMaxVals = 21;
constexpr int kRefitRedFloats = kRefitMaxVals * kSumTRow + 24;     // rows of 64 + 4 floats, then the totals

// (lane: the caller's lane index -- passed in so that a kernel can hand over an opaque copy, amis_forward_mfma.hip)
template <int NV>
PNP_FN void wave_sum_t(float (&v)[NV], float* lds, int lane) {
  static_assert(NV <= kRefitMaxVals, "scratch rows");
  float* tot = lds + kRefitMaxVals * kSumTRow;
#pragma unroll
  for (int i = 0; i < NV; ++i) lds[i * kSumTRow + lane] = v[i];
  wave_lds_fence();
#pragma unroll
  for (int base = 0; base < NV; base += 16) {
    const int row = base + (lane >> 2);
    float part = 0.f;
    if (row < NV) {
      const float4* r = reinterpret_cast<const float4*>(lds + row * kSumTRow + 16 * (lane & 3));
      const float4 a = r[0], b = r[1], c = r[2], d = r[3];
      part = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
    }
    part = quad_sum(part);
    if (row < NV && (lane & 3) == 0) tot[row] = part;
  }
  wave_lds_fence();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = tot[i];
  wave_lds_fence();
}

// the refit's cross-lane sums: transposed through `scratch`, or (no scratch: the all-VALU forward kernel) wave_sum chains
template <int NV>
PNP_FN void refit_sum(float (&v)[NV], float* scratch, int lane) {
  if (scratch != nullptr) {
    wave_sum_t<NV>(v, scratch, lane);
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  }
}

// Translation (3x3) and rotation (4x4, ACG) factors of a proposal fitted by TWO lanes in lockstep: lane 0 takes the rotation
// shape matrix, lane 1 the translation covariance padded to 4x4 with a unit pivot -- the same instructions serve both, so the
// translation fit rides for free on the ~250 fp64 instructions of the rotation fit (it used to follow it on the same lane).
// Rotation (epropnp.py:301-302,341-342): rot_cov (4x4 SPD, trace ~ 1) + det^(1/4) * dispersion * I -> Cholesky -> rec[16..36],
// identity factor when not SPD.  Translation: the same operations per matrix as fit_translation -- lane 1 has dispersion 0 (its diagonal
// shift is exactly 0, the second factorisation repeats the first) and a unit last pivot (log 1 = 0 in the normaliser).
// A: this lane's 4x4 matrix (lower triangle read); which = 0 rotation / 1 translation; fallback: diagonal used when A is not SPD.
PNP_FIT_FN void fit_factor_pair(fit_t (&A)[4][4], int which, float dispersion, const float (&fallback)[4], float* rec) {
  // det A by 2x2 minors of rows (0,1) and (2,3) (fp64: cancellation costs cond * 1e-16): 30 operations and no copy of A,
  // where a first Cholesky factorisation for its pivots cost 75 and ten more live doubles
  const fit_t m01 = A[0][0] * A[1][1] - A[0][1] * A[1][0], m02 = A[0][0] * A[1][2] - A[0][2] * A[1][0],
               m03 = A[0][0] * A[1][3] - A[0][3] * A[1][0], m12 = A[0][1] * A[1][2] - A[0][2] * A[1][1],
               m13 = A[0][1] * A[1][3] - A[0][3] * A[1][1], m23 = A[0][2] * A[1][3] - A[0][3] * A[1][2];
  const fit_t n01 = A[2][0] * A[3][1] - A[2][1] * A[3][0], n02 = A[2][0] * A[3][2] - A[2][2] * A[3][0],
               n03 = A[2][0] * A[3][3] - A[2][3] * A[3][0], n12 = A[2][1] * A[3][2] - A[2][2] * A[3][1],
               n13 = A[2][1] * A[3][3] - A[2][3] * A[3][1], n23 = A[2][2] * A[3][3] - A[2][3] * A[3][2];
  const float det = (float)(m01 * n23 - m02 * n13 + m03 * n12 + m12 * n03 - m13 * n02 + m23 * n01);
  // det^(1/4) * dispersion; a negative determinant (indefinite matrix) gives NaN here as torch.det(...) ** 0.25 does in the
  // reference, the factorisation below then fails and the factor falls back to its default
  const fit_t add = (which == 0) ? (fit_t)(sqrtf(sqrtf(det)) * dispersion) : fit_t(0);
  fit_t invd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) A[i][i] += add;
  const bool ok = cholesky<4, fit_t>(A, invd) && (add == add);
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) A[i][j] = (i == j) ? (fit_t)fallback[i] : fit_t(0);
      invd[i] = fit_t(1) / (fit_t)fallback[i];
    }
  }
  fit_t Li[4][4];
  tri_inverse<4, fit_t>(A, invd, Li);
  float sl = 0.f;
  const int oL = which ? 3 : 16, oLi = which ? 9 : 26;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sl += fast_log((float)A[i][i]);
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      if (which == 0 || i < 3) {
        rec[oL + tri(i, j)] = (float)A[i][j];
        rec[oLi + tri(i, j)] = (float)Li[i][j];
      }
    }
  }
  rec[which ? 15 : 36] = which ? student_t3_log_norm(sl) : sl;
  rec[which ? 37 : 38] = ok ? 0.f : 1.f;
}

// proposal #0 from the Laplace approximation at the LM solution.  6-DoF: called by lanes 0 AND 1 of a wave (`which` = lane):
// both run the same instructions, lane 1's translation factor comes out of the rotation fit's instruction stream
// (fit_factor_pair).  4-DoF: lane 0 only.
template <int DOF>
PNP_FIT_FN void initial_fit(const float* pose_opt, const float* cov, float eps, float dispersion, float* rec, int which = 0) {
  if (which == 0) {
    rec[0] = pose_opt[0]; rec[1] = pose_opt[1]; rec[2] = pose_opt[2];
    rec[39] = 0.f;
  }
  if (DOF == 4) {
    rec[37] = rec[38] = 0.f;
    fit_t Ct[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Ct[i][j] = (fit_t)cov[i * DOF + j];
    const float dflt[3] = {1.0f, 1.0f, 4.0f};
    fit_translation(Ct, dflt, rec);
    rec[16] = pose_opt[3];
    const float kappa = 0.33f / fmaxf(cov[3 * 4 + 3], eps);
    rec[17] = kappa;
    rec[18] = log_i0(kappa);
#pragma unroll
    for (int i = 19; i < 37; ++i) rec[i] = 0.f;      // unused in the 4-DoF record: the records are an output, no stale LDS in them
  } else {
    fit_t Cr[3][3], Ci[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Cr[i][j] = (fit_t)cov[(3 + i) * 6 + 3 + j];
    fit_t invd4[4];
    // the reference inverts this block with LU (torch.inverse, epropnp.py:297): a true inverse also when the block is
    // indefinite.  Adjugate / determinant in fp64 does the same (NaN in -> NaN out -> identity fallback below).
    {
      const fit_t c00 = Cr[1][1] * Cr[2][2] - Cr[1][2] * Cr[2][1], c01 = Cr[1][2] * Cr[2][0] - Cr[1][0] * Cr[2][2],
                   c02 = Cr[1][0] * Cr[2][1] - Cr[1][1] * Cr[2][0];
      const fit_t idet = fit_t(1) / (Cr[0][0] * c00 + Cr[0][1] * c01 + Cr[0][2] * c02);
      Ci[0][0] = c00 * idet; Ci[1][0] = c01 * idet; Ci[2][0] = c02 * idet;
      Ci[0][1] = (Cr[0][2] * Cr[2][1] - Cr[0][1] * Cr[2][2]) * idet;
      Ci[1][1] = (Cr[0][0] * Cr[2][2] - Cr[0][2] * Cr[2][0]) * idet;
      Ci[2][1] = (Cr[0][1] * Cr[2][0] - Cr[0][0] * Cr[2][1]) * idet;
      Ci[0][2] = (Cr[0][1] * Cr[1][2] - Cr[0][2] * Cr[1][1]) * idet;
      Ci[1][2] = (Cr[0][2] * Cr[1][0] - Cr[0][0] * Cr[1][2]) * idet;
      Ci[2][2] = (Cr[0][0] * Cr[1][1] - Cr[0][1] * Cr[1][0]) * idet;
    }
    const fit_t w = pose_opt[3], qi = pose_opt[4], qj = pose_opt[5], qk = pose_opt[6];
    const fit_t T[4][3] = {{qi, qj, qk}, {-w, -qk, qj}, {qk, -w, -qi}, {-qj, qi, -w}};   // camera.py:158-165
    fit_t TC[4][3], A[4][4], Ai[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) TC[i][j] = T[i][0] * Ci[0][j] + T[i][1] * Ci[1][j] + T[i][2] * Ci[2][j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        A[i][j] = TC[i][0] * T[j][0] + TC[i][1] * T[j][1] + TC[i][2] * T[j][2] + ((i == j) ? fit_t(1) : fit_t(0));
    // A = I + (rank <= 3) always has the eigenvalue 1, so a non-SPD A is indefinite, and so are its inverse and any
    // rescaling of it: the reference's Cholesky (epropnp.py:302) fails and falls back to I.  Same outcome here.
    const bool a_spd = spd_inverse<4, fit_t>(A, invd4, Ai);
    if (!a_spd) Ai[0][0] = -fit_t(1);
    const fit_t itr = fit_t(1) / (Ai[0][0] + Ai[1][1] + Ai[2][2] + Ai[3][3]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Ai[i][j] *= itr;
    if (which == 1) {      // this lane factors the translation covariance instead (padded with a unit pivot)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Ai[i][j] = (i < 3 && j < 3) ? (fit_t)cov[i * DOF + j] : ((i == j) ? fit_t(1) : fit_t(0));
    }
    const float dflt[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    fit_factor_pair(Ai, which, dispersion, dflt, rec);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Which samples the backward may skip.  sum_j a_j dc_j/dx is a sum of S terms whose weights a_j (softmax of the AMIS
// log-weights times the upstream gradient) span tens of orders of magnitude; the per-pair factor dc/dx is bounded
// (Huber), so terms whose TOTAL weight is below `eps` of the object's total |weight| move the result by less than eps
// relative -- at the default eps = 2^-24 that is one fp32 rounding of the sum itself.  The threshold is the largest
// power-of-two fraction of max|a| for which the dropped mass stays within that budget.  eps = 0 keeps every non-zero sample
// (exact).  Returns t: drop |a| <= t.
// Round 5: the 64 power-of-two bins hold their mass as 64-bit INTEGERS in units of max|a| * 2^-40, added with LDS atomics --
// integer sums do not depend on the order of the additions, so every thread bins its own samples and the result stays
// bit-reproducible; the suffix sums and the cut are taken by one wave, lane = bin.  (Until round 4 every wave walked through a
// quarter of the samples with all its lanes, one sample per trip, and lane 0 scanned the bins one dependent load at a time:
// 14 % of the backward's lifetime at 32 x 4096 points, 41 % at 32 x 512, 20 % at the detection shape --
// profiles/r05_phase_shares.txt.)  A sample below 2^-40 of the maximum counts as mass 0: S such samples are 2^-31 of the
// budget's unit at S = 512.
// `wabs(m)` = |a_m| for m < S; hist: LDS, 8-byte aligned, kDropHistFloats floats.  Ends on a barrier.
constexpr int kDropHistFloats = 2 * 66;
// Out of line (one call per workgroup).  History: inlined, this function changed the register allocation and schedule of the sweep
// loop behind it, and that build returned wrong gradients for 1-2 % of the points whenever two waves shared a SIMD.  The cause turned
// out to be packed fp32 instructions of one op_sel shape meeting a neighbour's bf16 MFMA (profiles/r05_pk_opsel_erratum.txt) -- which
// instructions the SLP vectoriser formed, and where they fell relative to the MFMAs, is what the inlining changed.  The backward is
// compiled without the vectoriser now (build.py); the call stays because it costs nothing measurable.
template <class W>
__device__ __attribute__((noinline)) float mass_drop_threshold(W&& wabs, int S, float amax, float eps, float* hist) {
  unsigned long long* bins = reinterpret_cast<unsigned long long*>(hist);      // [64] masses, [64] total, [65] threshold (float bits)
  const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = lane_id();
  const bool usable = (amax > 0.f) && (amax < INFINITY);          // all-zero / non-finite weights: drop nothing
  const float inv = usable ? 1.0f / amax : 0.f;
  if (tid < 64) bins[tid] = 0ull;
  __syncthreads();
  for (int m = tid; m < S; m += T) {
    const float rel = fminf(wabs(m) * inv, 1.0f);                 // (a non-finite weight among finite ones: NaN -> 1.0f)
    // bin k holds 2^-(k+1) < rel <= 2^-k (k = 0..62); bin 63 holds the rest, zeros included
    int k = (rel > 0.f) ? (int)(-__builtin_amdgcn_logf(rel)) : 63;
    k = min(max(k, 0), 63);
    const unsigned long long q = (unsigned long long)(rel * 0x1p40f);
    if (q != 0ull) atomicAdd(&bins[k], q);
  }
  __syncthreads();
  if (tid < 64) {       // wave 0, lane = bin: c = mass of bins lane..63; the cut is the lowest bin whose suffix stays within the budget
    unsigned long long c = 0ull;
    for (int j = 63; j >= 0; --j) c += (j >= lane) ? bins[j] : 0ull;
    if (lane == 0) bins[64] = c;
    wave_lds_fence();
    const unsigned long long total = bins[64];
    const unsigned long long budget = (unsigned long long)((double)total * (double)eps);
    const unsigned kmin = wave_min_u32((c <= budget) ? (unsigned)lane : 64u);
    if (lane == 0) hist[2 * 65] = (usable && kmin < 64u) ? ldexpf(amax, -(int)kmin) : 0.f;
  }
  __syncthreads();
  const float t = hist[2 * 65];
  __syncthreads();
  return t;
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
// Same bounded procedure (and the same accept / reject decisions) as oracle/epropnp_oracle.py:vm_sample_bounded, which
// follows numpy's legacy sampler:  z = cos(pi u1), f = (1 + r z) / (r + z), c = kappa (r - f),
// accept iff c (2 - c) > u2 or log(c / u2) + 1 - c >= 0, result = +-acos(f).
// Evaluated in a cancellation-free form so that fp32 suffices even for kappa ~ 1e4 (sharp posteriors):
//   r - f = (r^2 - 1) / (r + z),  r + z = (r - 1) + 2 cos^2(pi u1 / 2),  (1 -+ f) / 2 = (r -+ 1) {sin,cos}^2(pi u1 / 2) / (r + z)
// with (r - 1), (r^2 - 1) formed once per proposal in fp64.  Decisions whose margin is within 1e-4 (1 + c) of zero are
// re-evaluated in fp64 (a few per 10^4 attempts), so the outcome is that of the fp64 procedure.
// `next(a, u1, u2, u3)` supplies the three uniforms of attempt a (from injected noise or Philox): no per-lane array.
PNP_FN void sincos_half_pi(float u, float& s, float& c) {      // sin, cos of (pi / 2) u,  u in [0, 1]
  s = __builtin_amdgcn_sinf(0.25f * u);      // hardware argument is in revolutions
  c = __builtin_amdgcn_cosf(0.25f * u);
}

PNP_FN void sincos_rad(float x, float& s, float& c) {          // hardware sin / cos (argument in revolutions), |x| <~ 2 pi
  const float rev = x * 0.15915494309189535f;
  s = __builtin_amdgcn_sinf(rev);
  c = __builtin_amdgcn_cosf(rev);
}

template <class Uniforms>
PNP_FN float vm_sample_bounded(float loc, float kappa, Uniforms next) {
  const double k = fmax((double)kappa, 1e-12);
  const double tau = 1.0 + sqrt(1.0 + 4.0 * k * k);
  const double rho = (tau - sqrt(2.0 * tau)) / (2.0 * k);
  const double r = (k < 1e-5) ? (1.0 / k + k) : (1.0 + rho * rho) / (2.0 * rho);
  const double rm1 = (k < 1e-5) ? (r - 1.0) : (1.0 - rho) * (1.0 - rho) / (2.0 * rho);
  const double kq = k * rm1 * (r + 1.0);                  // kappa (r^2 - 1)
  const float kqf = (float)kq, rm1f = (float)rm1, srm1 = (float)sqrt(rm1), srp1 = (float)sqrt(r + 1.0);
  float x = 0.f;
  bool done = false;
  for (int a = 0; a < kVmTries; ++a) {
    // every active lane of the wave has accepted: the remaining attempts could not change anything (the acceptance
    // rate of Best-Fisher is >= 0.66, so a full wave is through after ~4-5 attempts instead of 16)
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
    float u1, u2, u3;
    next(a, u1, u2, u3);
    float sh, ch;
    sincos_half_pi(u1, sh, ch);
    const float den = fmaf(2.0f * ch, ch, rm1f);          // r + z > 0
    const float c = kqf / den;
    const float m1 = c * (2.0f - c) - u2;
    const float m2 = fast_log(c / fmaxf(u2, 1e-30f)) + 1.0f - c;      // margins within 1e-4 (1 + c) go to fp64 below
    const float tol = 1e-4f * (1.0f + c);
    bool acc = (m1 > 0.0f) || (m2 >= 0.0f);
    const bool clear = (u2 >= 1e-30f) && ((m1 > tol) || (m2 > tol) || ((m1 < -tol) && (m2 < -tol)));
    if (!clear) {                                          // borderline: decide in fp64
      const double chd = cos(1.5707963267948966 * (double)u1);
      const double cd = kq / (rm1 + 2.0 * chd * chd);
      acc = ((cd * (2.0 - cd) - (double)u2) > 0.0) || ((log(cd / fmax((double)u2, 1e-300)) + 1.0 - cd) >= 0.0);
    }
    if (!done && (acc || a == kVmTries - 1)) {
      // acos(f) = 2 atan2(sqrt((1 - f) / 2), sqrt((1 + f) / 2)); the common factor 1 / sqrt(r + z) cancels
      x = ((u3 - 0.5f >= 0.0f) ? 2.0f : -2.0f) * atan2f(srm1 * sh, srp1 * ch);
    }
    done = done || acc;
  }
  float y = x + loc + 3.14159265358979f;
  y = y - 6.283185307179586f * floorf(y * 0.15915494309189535f);
  return y - 3.14159265358979f;
}


// A sample's pose -> the (S, B, pose_len) output: 28 (16) contiguous bytes that are only dword-aligned.  Two (one) 16-byte
// stores with 4-byte alignment instead of seven (four) dword stores per lane (global memory takes unaligned wide accesses); the
// two of a 7-vector overlap in element 3, which the lane writes twice with the same value.
template <int PL>
PNP_FN void store_pose(float* dst, const float (&ps)[PL]) {
  typedef float f4_a4 __attribute__((vector_size(16), aligned(4)));
  const f4_a4 head = {ps[0], ps[1], ps[2], ps[3]};
  *reinterpret_cast<f4_a4*>(dst) = head;
  if (PL == 7) {
    const f4_a4 tail = {ps[3], ps[PL > 4 ? 4 : 0], ps[PL > 5 ? 5 : 0], ps[PL > 6 ? 6 : 0]};
    *reinterpret_cast<f4_a4*>(dst + 3) = tail;
  }
}

// Per-thread view of one object's LDS-resident sampler state (shared by the VALU and the MFMA forward kernels).
struct AmisCtx {
  float* ptab;    // [s_pad][12]  x|y|z rows of (K R | K t) of the current iteration's samples (A operands / broadcast rows)
  float* smp;     // [PL][S]      pose samples, SoA
  float* cst;     // [S]          cost of each sample
  float* mixl;    // [S]          log sum_j q_j(sample)
  float* lgw;     // [S]          log weight
  float* cpart;   // [WP][s]      partial costs of the current iteration (one row per point slice)
  float* prop;    // [K][kPropStride] fitted proposals
  float* red;     // [256]        block-reduction scratch
  float* nzb;     // [s][8] base noise of the NEXT draw, generated ahead by the idle waves (nullptr: drawn inline)
  float* rred;    // [kRefitRedFloats] scratch of the refit's single-wave transposed reductions (nullptr: DPP / readlane chains)
  int S, K, s, T, tid, b;
  int cstride;    // row stride of cpart (s, or s rounded up to 16 for the MFMA kernel)
};

// Base draws of sample m of object b: 3 normals + Chi2(3) for the Student-t translation, 4 normals for the ACG
// rotation (6-DoF).  Philox4x32-10 counter (b, m, offset, q), Box-Muller.  They do not depend on the fitted proposal,
// which is what lets the otherwise idle waves produce them while one lane runs the fp64 proposal fit.
template <int DOF>
PNP_FN void base_noise(const AmisParams& a, int b, int m, float (&out)[8]) {
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const uint32_t c2 = (uint32_t)a.offset, c3base = (uint32_t)(a.offset >> 32) * 64u;
  float nrm[12];
#pragma unroll
  for (int q = 0; q < (DOF == 6 ? 3 : 2); ++q) {
    const Philox4 r = philox4x32_10((uint32_t)b, (uint32_t)m, c2, c3base + q, k0, k1);
    box_muller(r.v[0], r.v[1], nrm[q * 4], nrm[q * 4 + 1]);
    box_muller(r.v[2], r.v[3], nrm[q * 4 + 2], nrm[q * 4 + 3]);
  }
  out[0] = nrm[0]; out[1] = nrm[1]; out[2] = nrm[2];
  out[3] = nrm[3] * nrm[3] + nrm[4] * nrm[4] + nrm[5] * nrm[5];   // Chi2(3)
  if (DOF == 6) {
    out[4] = nrm[6]; out[5] = nrm[7]; out[6] = nrm[8]; out[7] = nrm[9];
  } else {
    out[4] = out[5] = out[6] = out[7] = 0.f;
  }
}

// threads [first, T) of the workgroup fill cx.nzb with the base noise of iteration `it`
template <int DOF>
PNP_FN void amis_base_noise(const AmisCtx& cx, const AmisParams& a, int it, int first) {
  if (cx.nzb == nullptr || cx.tid < first) return;
  for (int n = cx.tid - first; n < cx.s; n += cx.T - first) {
    float nz8[8];
    base_noise<DOF>(a, cx.b, it * cx.s + n, nz8);
    reinterpret_cast<float4*>(cx.nzb)[2 * n] = make_float4(nz8[0], nz8[1], nz8[2], nz8[3]);
    reinterpret_cast<float4*>(cx.nzb)[2 * n + 1] = make_float4(nz8[4], nz8[5], nz8[6], nz8[7]);
  }
}

// ---------------- 1. draw s samples from proposal `it` (lane = sample) ----------------
// n0, cnt: the iteration's samples [n0, n0 + cnt) are drawn, their pose-table rows are 0 .. cnt-1 (cnt < 0: all s samples;
// the forward tiles an iteration whose pose table would not fit LDS)
template <int DOF>
PNP_FN void amis_draw(const AmisCtx& cx, const Problem& p, const AmisParams& a, int it, const float (&Kc)[9],
                      const float* __restrict__ noise, float* __restrict__ pose_samples, int n0 = 0, int cnt = -1) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NZ = (DOF == 6) ? 8 : 4 + 3 * kVmTries;
  float* ptab = cx.ptab; float* smp = cx.smp;
  const int S = cx.S, K = cx.K, s = cx.s, T = cx.T, tid = cx.tid, b = cx.b;
  const float* rec = cx.prop + it * kPropStride;
  // The pre-generated base noise may share LDS with the pose table this function fills (the two are never live at the
  // same time except here): every lane takes its noise into registers before any lane writes a pose row.
  const bool nz_aliased = (cx.nzb != nullptr) && (cx.nzb == cx.ptab);
  float4 nz_a = make_float4(0.f, 0.f, 0.f, 0.f), nz_b = nz_a;
  if (nz_aliased) {          // launcher guarantees s <= T in this mode: one sample per lane
    if (tid < s) {
      nz_a = reinterpret_cast<const float4*>(cx.nzb)[2 * tid];
      nz_b = reinterpret_cast<const float4*>(cx.nzb)[2 * tid + 1];
    }
    __syncthreads();
  }
  const int n_end = (cnt < 0) ? s : n0 + cnt;
  for (int n = n0 + tid; n < n_end; n += T) {
    const int m = it * s + n;
    float z[3], chi2, g[4];
    if (noise != nullptr) {
      const float* nz = noise + (((size_t)b * K + it) * s + n) * NZ;
      z[0] = nz[0]; z[1] = nz[1]; z[2] = nz[2]; chi2 = nz[3];
      if (DOF == 6) {
        g[0] = nz[4]; g[1] = nz[5]; g[2] = nz[6]; g[3] = nz[7];
      }
    } else if (cx.nzb != nullptr) {      // generated ahead of time by amis_base_noise
      const float4 n0 = nz_aliased ? nz_a : reinterpret_cast<const float4*>(cx.nzb)[2 * n];
      const float4 n1 = nz_aliased ? nz_b : reinterpret_cast<const float4*>(cx.nzb)[2 * n + 1];
      z[0] = n0.x; z[1] = n0.y; z[2] = n0.z; chi2 = n0.w;
      g[0] = n1.x; g[1] = n1.y; g[2] = n1.z; g[3] = n1.w;
    } else {
      float nz8[8];
      base_noise<DOF>(a, b, m, nz8);
      z[0] = nz8[0]; z[1] = nz8[1]; z[2] = nz8[2]; chi2 = nz8[3];
      g[0] = nz8[4]; g[1] = nz8[5]; g[2] = nz8[6]; g[3] = nz8[7];
    }
    float ps[PL];
    // translation: mode + L_t (z * rsqrt(chi2 / 3))
    const float sc = fast_rsqrt(chi2 * (1.0f / 3.0f));
    const float y0 = z[0] * sc, y1 = z[1] * sc, y2 = z[2] * sc;
    ps[0] = rec[0] + rec[3] * y0;
    ps[1] = rec[1] + (rec[4] * y0 + rec[5] * y1);
    ps[2] = rec[2] + (rec[6] * y0 + rec[7] * y1 + rec[8] * y2);
    PNP_REFIT_CLOCK(rot_t0_);
    if (DOF == 6) {   // ACG: L_r g / |L_r g|   (distributions.py:42-52)
      const float v0 = rec[16] * g[0];
      const float v1 = rec[17] * g[0] + rec[18] * g[1];
      const float v2 = rec[19] * g[0] + rec[20] * g[1] + rec[21] * g[2];
      const float v3 = rec[22] * g[0] + rec[23] * g[1] + rec[24] * g[2] + rec[25] * g[3];
      const float n2 = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
      if (n2 < 1e-12f) {            // |L g| < 1e-6
        ps[3] = 1.f; ps[4] = 0.f; ps[5] = 0.f; ps[6] = 0.f;
      } else {
        const float inr = rsqrt_newton(n2);      // unit norm to ~1e-7
        ps[3] = v0 * inr; ps[4] = v1 * inr; ps[5] = v2 * inr; ps[6] = v3 * inr;
      }
    } else {          // first round(0.25 s) samples uniform, the rest von Mises  (distributions.py:65-71)
      const int n_u = (int)rintf(0.25f * (float)s);
      // uniforms of attempt `att`: injected (row layout [z, chi2, 16 x (u1,u2,u3)]) or one Philox block per attempt
      auto uniforms = [&](int att, float& u1, float& u2, float& u3) {
        if (noise != nullptr) {
          const float* nz = noise + (((size_t)b * K + it) * s + n) * NZ + 4 + 3 * att;
          u1 = nz[0]; u2 = nz[1]; u3 = nz[2];
        } else {
          const Philox4 r = philox4x32_10((uint32_t)b, (uint32_t)m, (uint32_t)a.offset,
                                          (uint32_t)(a.offset >> 32) * 64u + 8u + (uint32_t)att, (uint32_t)a.seed,
                                          (uint32_t)(a.seed >> 32));
          u1 = (float)(r.v[0] >> 8) * (1.0f / 16777216.0f);
          u2 = (float)(r.v[1] >> 8) * (1.0f / 16777216.0f);
          u3 = (float)(r.v[2] >> 8) * (1.0f / 16777216.0f);
        }
      };
      if (n < n_u) {
        float u1, u2, u3;
        uniforms(0, u1, u2, u3);
        ps[3] = (u1 * 2.0f - 1.0f) * 3.14159265358979f;
      } else {
        ps[3] = vm_sample_bounded(rec[16], rec[17], uniforms);
      }
    }
    PNP_REFIT_ADD(3, rot_t0_);
#pragma unroll
    for (int i = 0; i < PL; ++i) smp[i * S + m] = ps[i];
    if (pose_samples != nullptr) store_pose<PL>(pose_samples + ((size_t)m * p.B + b) * PL, ps);
    {   // project_b operands of this sample -> LDS row (read back as broadcast by every lane of the sweep)
      float R[9], KR[9], Kt[3];
      pose_to_rot<DOF>(ps, R);
      if (pose_samples != nullptr && a.dn_samples != nullptr) {      // the sample in the caller's frame (AmisParams.dn_*)
        float pd[PL];
#pragma unroll
        for (int i = 0; i < PL; ++i) pd[i] = ps[i];
        shift_translation(pd, R, a.dn_offset[(size_t)b * 3], a.dn_offset[(size_t)b * 3 + 1], a.dn_offset[(size_t)b * 3 + 2], -1.0f);
        store_pose<PL>(a.dn_samples + ((size_t)m * p.B + b) * PL, pd);
      }
      compose_kr_kt(Kc, R, ps, KR, Kt);
      float4* row = reinterpret_cast<float4*>(ptab + 12 * (n - n0));
      // [x-row | y-row | z-row] of (K R | K t): doubles as the A operand of the MFMA sweep
      row[0] = make_float4(KR[0], KR[1], KR[2], Kt[0]);
      row[1] = make_float4(KR[3], KR[4], KR[5], Kt[1]);
      row[2] = make_float4(KR[6], KR[7], KR[8], Kt[2]);
    }
  }

}

// ---------------- 3+4. proposal densities, mixture, log-weights (lane = sample) ----------------
template <int DOF>
PNP_FN void amis_weights(const AmisCtx& cx, const AmisParams& a, int it, int WP) {
  constexpr int PL = PoseLen<DOF>::value;
  float* smp = cx.smp; float* cst = cx.cst; float* mixl = cx.mixl; float* lgw = cx.lgw; float* cpart = cx.cpart; float* prop = cx.prop;
  const int S = cx.S, s = cx.s, T = cx.T, tid = cx.tid;
  const float* rec = prop + it * kPropStride;
  const int M = (it + 1) * s;
  const float log_n = logf((float)(it + 1));
  for (int m = tid; m < M; m += T) {
    float ps[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) ps[i] = smp[i * S + m];
    float mix;
    if (m >= it * s) {   // new sample: every proposal so far
      if (WP > 0) {      // (WP == 0: the tiled sweep has stored the costs already)
        float c = cpart[m - it * s];
        for (int q = 1; q < WP; ++q) c += cpart[q * cx.cstride + (m - it * s)];
        cst[m] = c;
      }
      if (PNP_ABLATED(a, 4)) {
        mix = 0.f;
      } else {
        mix = proposal_logprob<DOF>(prop, ps);
        for (int j = 1; j <= it; ++j) mix = log_add_exp(mix, proposal_logprob<DOF>(prop + j * kPropStride, ps));
      }
    } else {             // old sample: add the new proposal's density
      mix = PNP_ABLATED(a, 4) ? 0.f : log_add_exp(mixl[m], proposal_logprob<DOF>(rec, ps));
    }
    mixl[m] = mix;
    lgw[m] = -cst[m] - (mix - log_n);
  }

}

// ---------------- 5. fit proposal it+1 to the weighted samples (epropnp.py:238-260 / :317-342) ---------
// Runs on wave 0 only: with <= a few hundred samples the moment passes are short, and a single wave needs no workgroup
// barriers (the other waves pre-generate the next draw's base noise and wait at the end).  What the phase costs is the
// number of instructions that one wave issues while three are parked (profiles/r02_tune_fwd_serial_phases.txt, r04): the
// cross-lane sums go through LDS transposed (wave_sum_t: ~35 instead of 231 VALU instructions for the 21 moments), the
// fixed point hands the samples the Cholesky factor L of Sigma instead of Sigma^-1 (|L^-1 q|^2 = q^T Sigma^-1 q by forward
// substitution: no triangular inverse and no L^-T L^-1 product on the fitting lane, 14 instead of 20 operations per sample,
// and a sum of squares instead of a cancelling quadratic form), the weights
// are divided by hardware reciprocals, and the translation factor is fitted by lane 1 inside the instruction stream of lane
// 0's rotation fit (fit_factor_pair).
template <int DOF>
PNP_FN void amis_refit(const AmisCtx& cx, const AmisParams& a, int it) {
  float* smp = cx.smp; float* lgw = cx.lgw; float* prop = cx.prop; float* red = cx.red;
  const int S = cx.S, s = cx.s;
  const float* rec = prop + it * kPropStride;
  const int M = (it + 1) * s;
  float* nrec = prop + (it + 1) * kPropStride;
  if (wave_id() != 0) {
    amis_base_noise<DOF>(cx, a, it + 1, 64);     // the next draw's Philox / Box-Muller work hides under the fit
    __syncthreads();
    return;
  }
  const int T = 64, tid = cx.tid;      // (wave 0: the thread index IS the lane index)
  PNP_REFIT_BEGIN();
  if (PNP_ABLATED(a, 2)) {
    for (int i = tid; i < kPropStride; i += T) nrec[i] = rec[i];
    __syncthreads();
    return;
  }
  // One pass over the samples gathers every first/second moment that does not depend on a matrix inverse:
  // Z = sum e, sum e d, sum e d d^T with d = t - (previous mode) as pivot (keeps the E[dd^T] - dd^T cancellation
  // benign), and -- since the ACG fixed point starts from Sigma = I -- the first maximum-likelihood step as well.
  float mx = -INFINITY;
  for (int m = tid; m < M; m += T) mx = fmaxf(mx, lgw[m]);
  mx = wave_max(mx);
  const float p0 = rec[0], p1 = rec[1], p2 = rec[2];
  if (DOF == 6) {
    float mom[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) mom[i] = 0.f;
    if (PNP_ABLATED(a, 8)) mom[0] = mom[4] = mom[6] = mom[9] = mom[10] = mom[12] = mom[15] = mom[19] = mom[20] = 1.f;
    else
    for (int m = tid; m < M; m += T) {
      const float e = fast_exp(lgw[m] - mx);
      const float d0 = smp[m] - p0, d1 = smp[S + m] - p1, d2 = smp[2 * S + m] - p2;
      mom[0] += e;
      mom[1] += e * d0; mom[2] += e * d1; mom[3] += e * d2;
      mom[4] += e * d0 * d0; mom[5] += e * d1 * d0; mom[6] += e * d1 * d1;
      mom[7] += e * d2 * d0; mom[8] += e * d2 * d1; mom[9] += e * d2 * d2;
      const float q0 = smp[3 * S + m], q1 = smp[4 * S + m], q2 = smp[5 * S + m], q3 = smp[6 * S + m];
      const float iw = e * fast_rcp(fmaxf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3, a.eps));    // M = q^T I q
      mom[10] += iw * q0 * q0;
      mom[11] += iw * q1 * q0; mom[12] += iw * q1 * q1;
      mom[13] += iw * q2 * q0; mom[14] += iw * q2 * q1; mom[15] += iw * q2 * q2;
      mom[16] += iw * q3 * q0; mom[17] += iw * q3 * q1; mom[18] += iw * q3 * q2; mom[19] += iw * q3 * q3;
      mom[20] += iw;
    }
    if (!PNP_ABLATED(a, 16)) refit_sum<21>(mom, cx.rred, tid);
    const float invZ = 1.0f / mom[0];
    const float dl0 = mom[1] * invZ, dl1 = mom[2] * invZ, dl2 = mom[3] * invZ;
    const float mu0 = p0 + dl0, mu1 = p1 + dl1, mu2 = p2 + dl2;
    float c6[6];
    c6[0] = mom[4] * invZ - dl0 * dl0; c6[1] = mom[5] * invZ - dl1 * dl0; c6[2] = mom[6] * invZ - dl1 * dl1;
    c6[3] = mom[7] * invZ - dl2 * dl0; c6[4] = mom[8] * invZ - dl2 * dl1; c6[5] = mom[9] * invZ - dl2 * dl2;
    float acc[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) acc[i] = mom[10 + i];
    PNP_REFIT_PHASE(0);
    float Lf[10], Ld[4];
    for (int r = 1; r < a.mle_iter; ++r) {
      // Cholesky factor of the previous fixed-point iterate Sigma = L L^T (one lane, fp64) -> the strict lower triangle and the
      // reciprocal pivots, broadcast through LDS: every sample then solves L y = q by forward substitution (10 operations,
      // as many as a product with L^-1 would take -- which the fitting lane therefore does not have to form)
      if (tid == 0) {
        fit_t Sg[4][4], invd[4];
        const fit_t inorm = fit_t(1) / (fit_t)acc[10];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            const fit_t v = (fit_t)acc[tri(i, j)] * inorm + ((i == j) ? (fit_t)a.eps : fit_t(0));
            Sg[i][j] = v;
            Sg[j][i] = v;
          }
        cholesky<4, fit_t>(Sg, invd);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          red[10 + i] = (float)invd[i];
#pragma unroll
          for (int j = 0; j < i; ++j) red[tri(i, j)] = (float)Sg[i][j];
        }
      }
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < 10; ++i) Lf[i] = red[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) Ld[i] = red[10 + i];
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < 11; ++i) acc[i] = 0.f;
      for (int m = tid; m < M; m += T) {
        const float e = fast_exp(lgw[m] - mx);
        const float q0 = smp[3 * S + m], q1 = smp[4 * S + m], q2 = smp[5 * S + m], q3 = smp[6 * S + m];
        const float y0 = q0 * Ld[0];
        const float y1 = fmaf(-Lf[tri(1, 0)], y0, q1) * Ld[1];
        const float y2 = fmaf(-Lf[tri(2, 1)], y1, fmaf(-Lf[tri(2, 0)], y0, q2)) * Ld[2];
        const float y3 = fmaf(-Lf[tri(3, 2)], y2, fmaf(-Lf[tri(3, 1)], y1, fmaf(-Lf[tri(3, 0)], y0, q3))) * Ld[3];
        const float Mq = fmaf(y3, y3, fmaf(y2, y2, fmaf(y1, y1, y0 * y0)));       // q^T Sigma^-1 q = |L^-1 q|^2
        const float iw = e * fast_rcp(fmaxf(Mq, a.eps));     // the reference normalises w first; the ratio below is scale-free
        acc[10] += iw;
        acc[0] += iw * q0 * q0;
        acc[1] += iw * q1 * q0; acc[2] += iw * q1 * q1;
        acc[3] += iw * q2 * q0; acc[4] += iw * q2 * q1; acc[5] += iw * q2 * q2;
        acc[6] += iw * q3 * q0; acc[7] += iw * q3 * q1; acc[8] += iw * q3 * q2; acc[9] += iw * q3 * q3;
      }
      refit_sum<11>(acc, cx.rred, tid);
    }
    PNP_REFIT_PHASE(1);
    if (tid < 2) {      // lane 0: rotation factor, lane 1: translation factor (the same instructions)
      if (tid == 0) {
        nrec[0] = mu0; nrec[1] = mu1; nrec[2] = mu2;
        nrec[39] = 0.f;
      }
      fit_t A4[4][4];
      const fit_t inorm = (a.mle_iter > 0) ? fit_t(1) / (fit_t)acc[10] : fit_t(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          fit_t v = (a.mle_iter > 0) ? (fit_t)acc[tri(i, j)] * inorm + ((i == j) ? (fit_t)a.eps : fit_t(0))
                                      : ((i == j) ? fit_t(1) : fit_t(0));
          if (tid == 1) v = (i < 3) ? (fit_t)c6[tri(i, j)] : ((i == j) ? fit_t(1) : fit_t(0));
          A4[i][j] = v;
          A4[j][i] = v;
        }
      const float dflt[4] = {1.f, 1.f, 1.f, 1.f};
      fit_factor_pair(A4, tid, a.dispersion, dflt, nrec);
    }
    PNP_REFIT_PHASE(2);
  } else {
    float mom[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) mom[i] = 0.f;
    for (int m = tid; m < M; m += T) {
      const float e = fast_exp(lgw[m] - mx);
      const float d0 = smp[m] - p0, d1 = smp[S + m] - p1, d2 = smp[2 * S + m] - p2;
      mom[0] += e;
      mom[1] += e * d0; mom[2] += e * d1; mom[3] += e * d2;
      mom[4] += e * d0 * d0; mom[5] += e * d1 * d0; mom[6] += e * d1 * d1;
      mom[7] += e * d2 * d0; mom[8] += e * d2 * d1; mom[9] += e * d2 * d2;
      const float yaw = smp[3 * S + m];
      float sy, cy;
      sincos_rad(yaw, sy, cy);
      mom[10] += e * sy;
      mom[11] += e * cy;
    }
    refit_sum<12>(mom, cx.rred, tid);
    const float invZ = 1.0f / mom[0];
    const float dl0 = mom[1] * invZ, dl1 = mom[2] * invZ, dl2 = mom[3] * invZ;
    const float mu0 = p0 + dl0, mu1 = p1 + dl1, mu2 = p2 + dl2;
    float c8[8];
    c8[0] = mom[4] * invZ - dl0 * dl0; c8[1] = mom[5] * invZ - dl1 * dl0; c8[2] = mom[6] * invZ - dl1 * dl1;
    c8[3] = mom[7] * invZ - dl2 * dl0; c8[4] = mom[8] * invZ - dl2 * dl1; c8[5] = mom[9] * invZ - dl2 * dl2;
    c8[6] = mom[10] * invZ;
    c8[7] = mom[11] * invZ;
    if (tid == 0) {
      nrec[0] = mu0; nrec[1] = mu1; nrec[2] = mu2;
      fit_t Ct[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          Ct[i][j] = (fit_t)c8[tri(i, j)];
          Ct[j][i] = (fit_t)c8[tri(i, j)];
        }
      const float dflt[3] = {1.f, 1.f, 4.f};
      nrec[38] = nrec[39] = 0.f;
      fit_translation(Ct, dflt, nrec);
      nrec[16] = atan2f(c8[6], c8[7]);
      const float r_sq = c8[6] * c8[6] + c8[7] * c8[7];
      const float kappa = 0.33f * fmaxf(sqrtf(r_sq), a.eps) * (2.f - r_sq) / fmaxf(1.f - r_sq, a.eps);
      nrec[17] = kappa;
      nrec[18] = log_i0(kappa);
#pragma unroll
      for (int i = 19; i < 37; ++i) nrec[i] = 0.f;
    }
  }
  __syncthreads();
}

}  // namespace pnp
