// pnp_math.h -- per-thread math of the EPro-PnP hot path (register-resident, no memory traffic).
//
// Follows the formulas of the reference (paths relative to the reference checkout):
//   rotation / tangent map ....... epropnp/common.py:21-64, epropnp/camera.py:145-165
//   projection + clamp ........... epropnp/camera.py:10-30,81-93
//   Jacobian + clip .............. epropnp/camera.py:100-143
//   Huber cost / rescaling ....... epropnp/cost_fun.py:8-20,45-84
//   pose update .................. epropnp/levenberg_marquardt.py:255-265
//   Student-t / ACG / vM-mixture . pyro MultivariateStudentT, epropnp/distributions.py:15-79,
//                                  torch/distributions/von_mises.py:24-89
// The small dense algebra (6x6 / 4x4 / 3x3) is fully unrolled so that everything lives in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define PNP_FN __device__ __forceinline__

namespace pnp {

// ---------------------------------------------------------------------------------------------------
// fast reciprocal / rsqrt for the VALU-bound AMIS sweeps (1 ulp hardware approximations);
// the LM path uses IEEE division and sqrtf.
// ---------------------------------------------------------------------------------------------------
// a product / a sum that the compiler must NOT contract into an fma with its neighbours (results pinned to the reference's
// separately rounded operations: the running norm_factor, the pixel grid of the dense pre-processing).  Contraction needs the
// `contract` flag on BOTH operations; the pragma keeps it off the one formed here (HIP's __fmul_rn / __fadd_rn are plain
// operators that -ffp-contract=fast still fuses).
PNP_FN float mul_unfused(float a, float b) {
#pragma clang fp contract(off)
  const float m = a * b;
  return m;
}
PNP_FN float add_unfused(float a, float b) {
#pragma clang fp contract(off)
  const float m = a + b;
  return m;
}

PNP_FN float fast_rcp(float x) {
  return __builtin_amdgcn_rcpf(x);
}
PNP_FN float fast_rsqrt(float x) {
  return __builtin_amdgcn_rsqf(x);
}
PNP_FN float fast_sqrt(float x) {
  return __builtin_amdgcn_sqrtf(x);
}

template <int DOF>
struct PoseLen {
  static constexpr int value = (DOF == 6) ? 7 : 4;
};

// ---------------------------------------------------------------------------------------------------
// rotations
// ---------------------------------------------------------------------------------------------------
// unit quaternion [w,i,j,k] -> row-major R; same association as the reference's no-grad branch:
// R = 2 (w [v]x + v v^T) + (w^2 - v.v) I, quaternion not normalised.
PNP_FN void quat_to_rot(float w, float x, float y, float z, float (&R)[9]) {
  const float dd = w * w - (x * x + y * y + z * z);
  R[0] = 2.f * (x * x) + dd;
  R[1] = 2.f * (x * y - w * z);
  R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z);
  R[4] = 2.f * (y * y) + dd;
  R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y);
  R[7] = 2.f * (y * z + w * x);
  R[8] = 2.f * (z * z) + dd;
}

PNP_FN void yaw_to_rot(float yaw, float (&R)[9]) {
  // hardware sin / cos (argument in revolutions, ~2^-21 absolute error: 3e-5 px at 800 px focal length) instead of the
  // ~80-instruction libm pair; every kernel builds R through this function, so forward and backward stay consistent
  const float rev = yaw * 0.15915494309189535f;
  const float c = __builtin_amdgcn_cosf(rev), s = __builtin_amdgcn_sinf(rev);
  R[0] = c;   R[1] = 0.f; R[2] = s;
  R[3] = 0.f; R[4] = 1.f; R[5] = 0.f;
  R[6] = -s;  R[7] = 0.f; R[8] = c;
}

template <int DOF>
PNP_FN void pose_to_rot(const float* pose, float (&R)[9]) {
  if (DOF == 6) quat_to_rot(pose[3], pose[4], pose[5], pose[6], R);
  else yaw_to_rot(pose[3], R);
}

// translation += sign * R o: ONE statement of the arithmetic for every kernel that moves a pose between the caller's frame and
// the centred one (the centring kernels, shift_poses*, the fused centre + cost launch, the AMIS forward's denormalised outputs),
// so that they agree to the bit whichever one runs.  Every product and sum is rounded on its own: left to -ffp-contract=fast the
// same source line came out as different fma chains depending on whether `sign` was a literal or an argument at the call site
// (literal: the outer add is re-associated into the chain), and the one-call forward then differed from the separate launches
// in the last bit of pose_init's cost (4-DoF, tests/test_fused_and_limits.py on the GPU).
PNP_FN void shift_translation(float* ps, const float (&R)[9], float ox, float oy, float oz, float sign) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float e = add_unfused(add_unfused(mul_unfused(R[3 * i], ox), mul_unfused(R[3 * i + 1], oy)), mul_unfused(R[3 * i + 2], oz));
    ps[i] = add_unfused(ps[i], mul_unfused(sign, e));
  }
}

// project_b operands: KR = K R, Kt = K t  (epropnp/camera.py:23-27)
PNP_FN void compose_kr_kt(const float (&K)[9], const float (&R)[9], const float* t, float (&KR)[9], float (&Kt)[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) KR[r * 3 + c] = K[r * 3] * R[c] + K[r * 3 + 1] * R[3 + c] + K[r * 3 + 2] * R[6 + c];
    Kt[r] = K[r * 3] * t[0] + K[r * 3 + 1] * t[1] + K[r * 3 + 2] * t[2];
  }
}

// pose_new = pose (+) step   (levenberg_marquardt.py:255-265; T(q) from camera.py:158-165)
template <int DOF>
PNP_FN void pose_add(const float* pose, const float* step, float* out) {
  out[0] = pose[0] + step[0];
  out[1] = pose[1] + step[1];
  out[2] = pose[2] + step[2];
  if (DOF == 4) {
    out[3] = pose[3] + step[3];
  } else {
    const float w = pose[3], i = pose[4], j = pose[5], k = pose[6];
    const float a = step[3], b = step[4], c = step[5];
    float q0 = w + (i * a + j * b + k * c);
    float q1 = i + (-w * a - k * b + j * c);
    float q2 = j + (k * a - w * b - i * c);
    float q3 = k + (-j * a + i * b - w * c);
    const float n = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
    out[3] = q0 / n; out[4] = q1 / n; out[5] = q2 / n; out[6] = q3 / n;
  }
}

// adjoint of pose_add w.r.t. the step (the pose itself is not differentiated, as in the reference's use):
//   6-DoF: u = q + T(q) d, q+ = u / |u|  =>  dL/du = (g_q - q+ (q+ . g_q)) / |u|,  dL/dd = T(q)^T dL/du
template <int DOF>
PNP_FN void pose_add_adjoint(const float* pose, const float* step, const float* g_out, float* g_step) {
  g_step[0] = g_out[0]; g_step[1] = g_out[1]; g_step[2] = g_out[2];
  if (DOF == 4) {
    g_step[3] = g_out[3];
  } else {
    const float w = pose[3], i = pose[4], j = pose[5], k = pose[6];
    const float a = step[3], b = step[4], c = step[5];
    const float u0 = w + (i * a + j * b + k * c);
    const float u1 = i + (-w * a - k * b + j * c);
    const float u2 = j + (k * a - w * b - i * c);
    const float u3 = k + (-j * a + i * b - w * c);
    const float n = fmaxf(sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + u3 * u3), 1e-12f), inv = 1.0f / n;
    const float p0 = u0 * inv, p1 = u1 * inv, p2 = u2 * inv, p3 = u3 * inv;
    const float dot = p0 * g_out[3] + p1 * g_out[4] + p2 * g_out[5] + p3 * g_out[6];
    const float h0 = (g_out[3] - p0 * dot) * inv, h1 = (g_out[4] - p1 * dot) * inv;
    const float h2 = (g_out[5] - p2 * dot) * inv, h3 = (g_out[6] - p3 * dot) * inv;
    g_step[3] = i * h0 - w * h1 + k * h2 - j * h3;       // columns of T(q): [i,-w,k,-j], [j,-k,-w,i], [k,j,-i,-w]
    g_step[4] = j * h0 - k * h1 - w * h2 + i * h3;
    g_step[5] = k * h0 + j * h1 - i * h2 - w * h3;
  }
}

// ---------------------------------------------------------------------------------------------------
// one 2D-3D correspondence, as kept in registers
// ---------------------------------------------------------------------------------------------------
struct Point {
  float X, Y, Z;    // x3d
  float u, v;       // x2d
  float wu, wv;     // w2d
};

struct Bounds {   // projection clamp; only used when the kernel is instantiated with BOUNDS = true
  float lbx, lby, ubx, uby;
};

// clamp(x, lo, hi) of camera.py:81-93 as ONE instruction: the median of three (v_med3_f32) is the clamp whenever lo <= hi,
// a NaN x comes out as lo exactly as fminf(fmaxf(x, lo), hi) gives it, and x < lo / x > hi return lo / hi themselves, so
// the equality tests of clip_jac still see the bound.  min + max are two half-rate instructions per coordinate: 8 of them per
// two point-poses were a fifth of the AMIS sweeps' VALU time under a projection clamp (profiles/r03_tune_clamp_med3.txt).
PNP_FN float clamp_lu(float x, float lo, float hi) {
  return __builtin_amdgcn_fmed3f(x, lo, hi);
}

// Huber cost of one point under pose (KR, Kt): cost-only path (project_b).  FAST selects the 1-ulp
// hardware rcp/sqrt (AMIS sweeps); otherwise IEEE division / sqrt (cost_init, evaluate_cost).
template <bool BOUNDS, bool FAST>
PNP_FN float point_cost(const Point& p, const float (&KR)[9], const float (&Kt)[3], float z_min, float delta,
                        const Bounds& bd) {
  const float hx = fmaf(KR[0], p.X, fmaf(KR[1], p.Y, fmaf(KR[2], p.Z, Kt[0])));
  const float hy = fmaf(KR[3], p.X, fmaf(KR[4], p.Y, fmaf(KR[5], p.Z, Kt[1])));
  const float hz = fmaf(KR[6], p.X, fmaf(KR[7], p.Y, fmaf(KR[8], p.Z, Kt[2])));
  const float z = fmaxf(hz, z_min);
  float px, py;
  if (FAST) {
    const float rz = fast_rcp(z);
    px = hx * rz;
    py = hy * rz;
  } else {
    px = hx / z;
    py = hy / z;
  }
  if (BOUNDS) {
    px = clamp_lu(px, bd.lbx, bd.ubx);
    py = clamp_lu(py, bd.lby, bd.uby);
  }
  const float rx = (px - p.u) * p.wu;
  const float ry = (py - p.v) * p.wv;
  const float s = fmaf(rx, rx, ry * ry);
  const float rho = FAST ? fast_sqrt(s) : sqrtf(s);
  // huber: rho <= delta ? rho^2/2 : delta*rho - delta^2/2  ==  m*(rho - m/2), m = min(rho, delta)
  const float m = fminf(rho, delta);
  return m * fmaf(-0.5f, m, rho);
}

// Force a (possibly wave-uniform, SGPR-resident) value into a vector register.
PNP_FN float to_vgpr(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

// Correspondence in the form the AMIS cost sweep consumes: the residual is one FMA,
//   r = (p - u) * w  ==  fma(p, w, -u*w).
struct SweepPoint {
  float X, Y, Z, wu, wv, cu, cv;
};
PNP_FN SweepPoint to_sweep_point(const Point& p) {
  SweepPoint q;
  q.X = p.X; q.Y = p.Y; q.Z = p.Z; q.wu = p.wu; q.wv = p.wv;
  q.cu = -p.u * p.wu;
  q.cv = -p.v * p.wv;
  return q;
}

// Huber cost of one point under pose (KR, Kt), all operands in VGPRs: 17 plain VALU + max + min + rcp + sqrt.
template <bool BOUNDS>
PNP_FN float sweep_cost(const SweepPoint& p, const float (&KR)[9], const float (&Kt)[3], float z_min, float delta,
                        const Bounds& bd) {
  const float hx = fmaf(KR[0], p.X, fmaf(KR[1], p.Y, fmaf(KR[2], p.Z, Kt[0])));
  const float hy = fmaf(KR[3], p.X, fmaf(KR[4], p.Y, fmaf(KR[5], p.Z, Kt[1])));
  const float hz = fmaf(KR[6], p.X, fmaf(KR[7], p.Y, fmaf(KR[8], p.Z, Kt[2])));
  const float rz = fast_rcp(fmaxf(hz, z_min));
  float px = hx * rz, py = hy * rz;
  if (BOUNDS) {
    px = clamp_lu(px, bd.lbx, bd.ubx);
    py = clamp_lu(py, bd.lby, bd.uby);
  }
  const float rx = fmaf(px, p.wu, p.cu);
  const float ry = fmaf(py, p.wv, p.cv);
  const float rho = fast_sqrt(fmaf(rx, rx, ry * ry));
  const float m = fminf(rho, delta);          // huber = m * (rho - m/2)
  return m * fmaf(-0.5f, m, rho);
}

// Exact-form Huber (same expression as cost_fun.py:8-12) for the non-fast paths.
PNP_FN float huber_exact(float rho, float delta) {
  return (rho <= delta) ? 0.5f * rho * rho : delta * rho - 0.5f * delta * delta;
}

// ---------------------------------------------------------------------------------------------------
// tiny dense linear algebra, fully unrolled
// ---------------------------------------------------------------------------------------------------
// 1/sqrt(d) without the long IEEE sqrt/divide sequences: hardware rsq (fp32, 1 ulp) + one Newton step in the working
// precision.  fp32: ~full accuracy.  fp64: the seed is good to ~1e-7 (fp32 rounding of d + 1 ulp of v_rsq_f32), one step
// squares that: ~1.5e-14 relative -- what the fp64 proposal fits need is freedom from cancellation, not the last two digits
// (their results are stored as fp32), and each further step is five dependent fp64 instructions on the one lane the rest of
// the workgroup waits for.
PNP_FN float rsqrt_newton(float d) {
  const float r = fast_rsqrt(d);
  return r * fmaf(-0.5f * d * r, r, 1.5f);
}
PNP_FN double rsqrt_newton(double d) {
  const double r = (double)fast_rsqrt((float)d);
  return r * (1.5 - 0.5 * d * r * r);
}

// Cholesky A = L L^T in place on the lower triangle; invd[j] = 1 / L[j][j].
// Returns false when a pivot is not positive/finite (that pivot is then replaced by 1).
template <int D, typename T>
PNP_FN bool cholesky(T (&A)[D][D], T (&invd)[D]) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    T d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    const bool good = (d > T(1e-37)) && (d < T(1e37));
    ok = ok && good;
    d = good ? d : T(1);
    const T r = rsqrt_newton(d);
    A[j][j] = d * r;
    invd[j] = r;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      T s = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
      A[i][j] = s * r;
    }
  }
  return ok;
}

// solve L L^T x = b in place (L lower, invd = 1/diag L)
template <int D, typename T>
PNP_FN void cholesky_solve(const T (&L)[D][D], const T (&invd)[D], T (&b)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    T s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i][k] * b[k];
    b[i] = s * invd[i];
  }
#pragma unroll
  for (int i = D - 1; i >= 0; --i) {
    T s = b[i];
#pragma unroll
    for (int k = i + 1; k < D; ++k) s -= L[k][i] * b[k];
    b[i] = s * invd[i];
  }
}

// inverse of a lower-triangular matrix (result lower-triangular, upper part zeroed)
template <int D, typename T>
PNP_FN void tri_inverse(const T (&L)[D][D], const T (&invd)[D], T (&Li)[D][D]) {
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if (i < j) {
        Li[i][j] = T(0);
      } else if (i == j) {
        Li[i][j] = invd[i];
      } else {
        T s = T(0);
#pragma unroll
        for (int k = j; k < i; ++k) s -= L[i][k] * Li[k][j];
        Li[i][j] = s * invd[i];
      }
    }
  }
}

// SPD inverse through Cholesky: A^-1 = L^-T L^-1.  A is overwritten by its Cholesky factor, invd = 1/diag L.
// Returns false when A is not positive definite (result then undefined).
template <int D, typename T>
PNP_FN bool spd_inverse(T (&A)[D][D], T (&invd)[D], T (&Ainv)[D][D]) {
  const bool ok = cholesky<D, T>(A, invd);
  T Li[D][D];
  tri_inverse<D, T>(A, invd, Li);
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      T s = T(0);
#pragma unroll
      for (int k = i; k < D; ++k) s += Li[k][i] * Li[k][j];
      Ainv[i][j] = s;
      Ainv[j][i] = s;
    }
  return ok;
}

// Jacobi-scaled Cholesky of a symmetric positive definite matrix H (fp32): with s_i = 1/sqrt(H_ii),
// A = diag(s) H diag(s) has unit diagonal and a condition number that no longer carries the metres-vs-radians
// scale gap of the pose parameterisation, so single precision suffices.  L L^T = A on return (lower).
template <int D>
struct ScaledFactor {
  float L[D][D];     // Cholesky factor of diag(s) H diag(s)
  float invd[D];     // 1 / diag L
  float s[D];        // 1 / sqrt(diag H)
};

template <int D>
PNP_FN bool scaled_cholesky(const float (&H)[D][D], ScaledFactor<D>& f) {
#pragma unroll
  for (int i = 0; i < D; ++i) f.s[i] = rsqrt_newton(fmaxf(H[i][i], 1e-30f));
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) f.L[i][j] = H[i][j] * (f.s[i] * f.s[j]);
  return cholesky<D, float>(f.L, f.invd);
}

// x = H^-1 b through the scaled factor:  x = s .* (A^-1 (s .* b))
template <int D>
PNP_FN void scaled_solve(const ScaledFactor<D>& f, float (&b)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) b[i] *= f.s[i];
  cholesky_solve<D, float>(f.L, f.invd, b);
#pragma unroll
  for (int i = 0; i < D; ++i) b[i] *= f.s[i];
}

// H^-1 = diag(s) A^-1 diag(s)
template <int D>
PNP_FN void scaled_inverse(const ScaledFactor<D>& f, float (&Hinv)[D][D]) {
  float Li[D][D];
  tri_inverse<D, float>(f.L, f.invd, Li);
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int k = i; k < D; ++k) acc += Li[k][i] * Li[k][j];
      acc *= f.s[i] * f.s[j];
      Hinv[i][j] = acc;
      Hinv[j][i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------
// proposal densities
// ---------------------------------------------------------------------------------------------------
// natural log / exp on the hardware base-2 units (v_log_f32 / v_exp_f32, ~1 ulp): the densities need ~1e-6 ABSOLUTE
// accuracy of a log-probability, which these give at a fifth of the instructions of the libm expansions
PNP_FN float fast_log(float x) {
  return 0.6931471805599453f * __builtin_amdgcn_logf(x);
}
PNP_FN float fast_exp(float x) {
  return __builtin_amdgcn_exp2f(1.4426950408889634f * x);
}
constexpr float kLogPi = 1.1447298858494002f;
constexpr float kLog2Pi = 1.8378770664093453f;

// Student-t (df = 3, n = 3) normaliser:  sum log diag L + 1.5 log 3 + 1.5 log pi + lgamma(1.5) - lgamma(3)
//   lgamma(1.5) = log(sqrt(pi)/2), lgamma(3) = log 2
PNP_FN float student_t3_log_norm(float sum_log_diag) {
  return sum_log_diag + 1.5f * 1.0986122886681098f + 1.5f * kLogPi + (-0.12078223763524522f) - 0.6931471805599453f;
}
PNP_FN float student_t3_logprob(float maha, float log_norm) {
  return -3.0f * fast_log(fmaf(maha, 1.0f / 3.0f, 1.0f)) - log_norm;
}

// ACG on S^3 (q = 4): -2 log(maha) - sum log diag L - log(2 pi^2)
PNP_FN float acg4_logprob(float maha, float sum_log_diag) {
  return -2.0f * fast_log(maha) - sum_log_diag - 2.9826069522587457f;
}

// log I0(x), polynomial of torch/distributions/von_mises.py:24-89 (Abramowitz & Stegun 9.8.1 / 9.8.2)
PNP_FN float log_i0(float x) {
  if (x < 3.75f) {
    float y = x / 3.75f;
    y = y * y;
    float r = 0.45813e-2f;
    r = 0.360768e-1f + y * r;
    r = 0.2659732f + y * r;
    r = 1.2067492f + y * r;
    r = 3.0899424f + y * r;
    r = 3.5156229f + y * r;
    r = 1.0f + y * r;
    return fast_log(r);
  }
  const float y = 3.75f / x;
  float r = 0.392377e-2f;
  r = -0.1647633e-1f + y * r;
  r = 0.2635537e-1f + y * r;
  r = -0.2057706e-1f + y * r;
  r = 0.916281e-2f + y * r;
  r = -0.157565e-2f + y * r;
  r = 0.225319e-2f + y * r;
  r = 0.1328592e-1f + y * r;
  r = 0.39894228f + y * r;
  return x - 0.5f * fast_log(x) + fast_log(r);
}

// 0.75 von Mises + 0.25 uniform on the circle (epropnp/distributions.py:74-79); log_i0k = log_i0(kappa)
PNP_FN float vm_mix_logprob(float x, float loc, float kappa, float log_i0k) {
  const float a = kappa * cosf(x - loc) - kLog2Pi - log_i0k + (-0.2876820724517809f);   // + log 0.75
  const float b = -3.224171427529236f;                                                  // log(0.25 / 2pi)
  const float mx = fmaxf(a, b);
  return mx + fast_log(1.0f + fast_exp(-fabsf(a - b)));
}

PNP_FN float log_add_exp(float a, float b) {
  const float mx = fmaxf(a, b);
  if (mx == -INFINITY) return -INFINITY;
  return mx + fast_log(1.0f + fast_exp(-fabsf(a - b)));
}

// ---------------------------------------------------------------------------------------------------
// counter-based RNG: Philox4x32-10 (Salmon et al., SC'11), Box-Muller normals
// ---------------------------------------------------------------------------------------------------
struct Philox4 {
  uint32_t v[4];
};

PNP_FN uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

PNP_FN Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 out;
  out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
  return out;
}

// uint32 -> uniform in (0, 1]
PNP_FN float u01(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }

// two independent N(0,1) from two uint32
// Hardware v_log / v_sin / v_cos (the latter take their argument in revolutions, exactly what Box-Muller wants):
// ~10 instructions instead of ~110 for the libm versions; absolute error ~1e-6, irrelevant for random draws.
// Exponential-race key of point n with weight w (weighted sampling without replacement: the n_pts smallest keys win),
// packed for integer min-reductions: a non-negative float orders like its bit pattern, and the low 9 mantissa bits
// carry the point index (n < 512; the 2^-14 relative perturbation of the key is immaterial to the sampling law, and ties
// break towards the lower index for free).  Zero-weight points get +inf | n: never picked before any positive weight.
constexpr unsigned kRaceIdxBits = 9, kRaceIdxMask = (1u << kRaceIdxBits) - 1, kRaceInf = 0x7F800000u;
// inv_w = 1 / w for w > 0, anything else (0, inf, NaN) for a point that must not be picked: the fused initialiser forms it once
// per point and object and reuses it for every proposal (one multiply per key where the division was ~10 instructions)
PNP_FN float race_inv_weight(float w) { return (w > 0.f) ? 1.0f / w : 0.f; }
PNP_FN unsigned race_key(uint32_t rnd, float inv_w, int n) {
  if (!(inv_w > 0.f) || inv_w == INFINITY) return kRaceInf | (unsigned)n;
  // -ln(u) / w up to the constant factor ln 2 (irrelevant to the order): hardware log2, one multiply
  const float k = fabsf(__builtin_amdgcn_logf(u01(rnd))) * inv_w;
  unsigned bits;
  memcpy(&bits, &k, sizeof(bits));       // bit cast (compiles to a move)
  return ((bits < kRaceInf ? bits : kRaceInf - 1u) & ~kRaceIdxMask) | (unsigned)n;      // (an overflowing key stays below "never")
}

PNP_FN void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float r = fast_sqrt(-1.3862943611198906f * __builtin_amdgcn_logf(u01(a)));   // -2 ln u = -2 ln2 log2 u
  const float rev = u01(b);
  n0 = r * __builtin_amdgcn_cosf(rev);
  n1 = r * __builtin_amdgcn_sinf(rev);
}

}  // namespace pnp
