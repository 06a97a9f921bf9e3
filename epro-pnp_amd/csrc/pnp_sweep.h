// pnp_sweep.h -- device-side view of one batch of PnP problems, point loading, and the per-point
// "evaluate with Jacobian -> normal equations" body shared by the sweep kernel and the fused LM kernel.
#pragma once
#include "pnp_math.h"
#include "wave_ops.h"

namespace pnp {

// by-value kernel argument (device pointers); mirrors epropnp_problem of include/epropnp_hip.h
struct Problem {
  const float* __restrict__ x3d;
  const float* __restrict__ x2d;
  const float* __restrict__ w2d;
  const float* __restrict__ cam;
  const float* __restrict__ lb;
  const float* __restrict__ ub;
  const float* __restrict__ delta;
  float z_min;
  int B, N;
  float huber_eps, inv_huber_eps;   // HuberPnPCost.eps (default 1e-10) and its reciprocal
  int* status;                      // optional int32[2]: [0] |= flags, [1] = min(object index); see epropnp_hip.h
  const float* delta_stats;         // optional (B,4): delta came from adaptive_delta on this w2d (epropnp_hip.h)
  float delta_relative;
};

// record a numerical event for the caller (no-op without a status word); rare by construction, so plain atomics.  System
// scope: the word may live in host memory (the library's default status word, pnp_host.h:default_status_word).
__device__ __forceinline__ void raise_status(const Problem& p, int flags, int b) {
  if (p.status != nullptr && flags != 0) {
    __hip_atomic_fetch_or(p.status, flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_min(p.status + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// XCD-aware object index: the dispatcher is observed to place workgroup g on XCD g % 8, so give each XCD
// a contiguous range of objects -- neighbouring objects then share an L2 and their partial-line output
// writes ((S,B,.) layouts put objects b, b+1 in the same cache line) merge there.  Speed only.
__device__ __forceinline__ int object_of_block(int B) {
  const int g = (int)blockIdx.x;
  const int per = (B + 7) >> 3;
  const int b = (g & 7) * per + (g >> 3);
  return b;   // may be >= B for the padded tail: caller must check
}
__host__ __device__ __forceinline__ int padded_object_grid(int B) { return ((B + 7) >> 3) << 3; }

typedef float pnp_f32x2 __attribute__((vector_size(8)));

// NT: non-temporal loads for kernels that stream an object's correspondences exactly once (the Jacobian sweep, the LM
// solve): the lines are not kept in L2 / the Infinity Cache on their way through (MI355X_MICROARCH.md "nt-weights").
template <bool NT = false>
__device__ __forceinline__ Point load_point(const Problem& p, int b, int n) {
  Point q;
  if (n < p.N) {
    const size_t i = (size_t)b * (size_t)p.N + (size_t)n;
    const float* a = p.x3d + i * 3;
    if (NT) {
      q.X = __builtin_nontemporal_load(a); q.Y = __builtin_nontemporal_load(a + 1); q.Z = __builtin_nontemporal_load(a + 2);
      const pnp_f32x2 u = __builtin_nontemporal_load(reinterpret_cast<const pnp_f32x2*>(p.x2d + i * 2));
      const pnp_f32x2 w = __builtin_nontemporal_load(reinterpret_cast<const pnp_f32x2*>(p.w2d + i * 2));
      q.u = u[0]; q.v = u[1]; q.wu = w[0]; q.wv = w[1];
      return q;
    }
    q.X = a[0]; q.Y = a[1]; q.Z = a[2];
    const float2 u = *reinterpret_cast<const float2*>(p.x2d + i * 2);
    const float2 w = *reinterpret_cast<const float2*>(p.w2d + i * 2);
    q.u = u.x; q.v = u.y; q.wu = w.x; q.wv = w.y;
  } else {   // padding: zero weight => zero cost, zero Jacobian, zero gradient
    q.X = q.Y = q.Z = q.u = q.v = q.wu = q.wv = 0.f;
  }
  return q;
}

// The same point, loaded UNCONDITIONALLY (index clamped to N - 1, the weights of a lane beyond N zeroed afterwards: zero cost,
// zero Jacobian), non-temporal.  For sweeps that issue all of a lane's loads up front and then work through the points in
// order: with the loads under per-point `n < N` branches the compiler cannot know how many of the later loads were issued,
// so the s_waitcnt in front of the FIRST point's arithmetic waits for all of them; unconditional loads get exact counts and
// point k's arithmetic runs underneath the loads of points k+1.. (together with sched_fence, wave_ops.h).  Requires N >= 1.
__device__ __forceinline__ Point load_point_streamed(const Problem& p, int b, int n) {
  Point q;
  const int nc = min(n, p.N - 1);
  const size_t i = (size_t)b * (size_t)p.N + (size_t)nc;
  const float* a = p.x3d + i * 3;
  q.X = __builtin_nontemporal_load(a); q.Y = __builtin_nontemporal_load(a + 1); q.Z = __builtin_nontemporal_load(a + 2);
  const pnp_f32x2 u = __builtin_nontemporal_load(reinterpret_cast<const pnp_f32x2*>(p.x2d + i * 2));
  const pnp_f32x2 w = __builtin_nontemporal_load(reinterpret_cast<const pnp_f32x2*>(p.w2d + i * 2));
  q.u = u[0]; q.v = u[1]; q.wu = w[0]; q.wv = w[1];
  return q;
}
// ... and the masking of a lane beyond N, applied where the point is consumed (not at the load: that would wait for it)
__device__ __forceinline__ void mask_point_beyond(Point& q, int n, int N) {
  if (n >= N) q.wu = q.wv = 0.f;
}

template <bool BOUNDS>
__device__ __forceinline__ void load_camera(const Problem& p, int b, float (&K)[9], Bounds& bd, float& delta) {
#pragma unroll
  for (int i = 0; i < 9; ++i) K[i] = p.cam[(size_t)b * 9 + i];
  if (BOUNDS) {
    bd.lbx = p.lb[(size_t)b * 2]; bd.lby = p.lb[(size_t)b * 2 + 1];
    bd.ubx = p.ub[(size_t)b * 2]; bd.uby = p.ub[(size_t)b * 2 + 1];
  } else {
    bd.lbx = bd.lby = -INFINITY;
    bd.ubx = bd.uby = INFINITY;
  }
  delta = p.delta[b];
}

template <int DOF>
struct NormalEq {
  static constexpr int NH = DOF * (DOF + 1) / 2;   // upper triangle of J^T J, row-major
  static constexpr int NV = NH + DOF + 1;          // + J^T r + cost
};

// One point's contribution to (J^T J, J^T r, cost) at pose (R, t): the Jacobian path
// project_a -> clamp -> Jacobian -> clip -> Huber rescale  (camera.py:10-18,81-143; cost_fun.py:45-84).
// 1/z, |r| and the Huber rescaling come from the hardware rcp / rsq seeds plus one Newton step each (<= 1 ulp, no
// IEEE divide/sqrt expansions: those were a quarter of this loop's instructions).  One reciprocal replaces the
// reference's eight divisions by z; gamma = sqrt(min(delta / rho, 1)) reuses 1/rho from the norm.
template <int DOF, bool BOUNDS>
PNP_FN void point_normal_eq(const Point& p, const float (&K)[9], const float (&R)[9], const float* t, float z_min,
                            float delta, float inv_eps, const Bounds& bd, bool clip, float (&acc)[NormalEq<DOF>::NV]) {
  const float xr0 = R[0] * p.X + R[1] * p.Y + R[2] * p.Z;
  const float xr1 = R[3] * p.X + R[4] * p.Y + R[5] * p.Z;
  const float xr2 = R[6] * p.X + R[7] * p.Y + R[8] * p.Z;
  const float c0 = xr0 + t[0], c1 = xr1 + t[1], c2 = xr2 + t[2];
  const float hx = c0 * K[0] + c1 * K[1] + c2 * K[2];
  const float hy = c0 * K[3] + c1 * K[4] + c2 * K[5];
  const float hz = c0 * K[6] + c1 * K[7] + c2 * K[8];
  const float z = fmaxf(hz, z_min);
  float rz = fast_rcp(z);
  rz = rz * fmaf(-z, rz, 2.0f);
  float px = hx * rz, py = hy * rz;
  if (BOUNDS) {
    px = clamp_lu(px, bd.lbx, bd.ubx);
    py = clamp_lu(py, bd.lby, bd.uby);
  }
  float J0[DOF], J1[DOF];
  J0[0] = K[0] * rz; J0[1] = K[1] * rz; J0[2] = (K[2] - px) * rz;
  J1[0] = K[3] * rz; J1[1] = K[4] * rz; J1[2] = (K[5] - py) * rz;
  if (DOF == 6) {
    const float ax = 2.f * xr0, ay = 2.f * xr1, az = 2.f * xr2;
    J0[3] = J0[1] * az - J0[2] * ay;
    J0[4] = J0[2] * ax - J0[0] * az;
    J0[5] = J0[0] * ay - J0[1] * ax;
    J1[3] = J1[1] * az - J1[2] * ay;
    J1[4] = J1[2] * ax - J1[0] * az;
    J1[5] = J1[0] * ay - J1[1] * ax;
  } else {
    J0[3] = J0[0] * xr2 - J0[2] * xr0;
    J1[3] = J1[0] * xr2 - J1[2] * xr0;
  }
  const float rx = (px - p.u) * p.wu;
  const float ry = (py - p.v) * p.wv;
  const float s2 = rx * rx + ry * ry;
  const float ir = fast_rsqrt(fmaxf(s2, 1e-36f));             // ~ 1 / rho
  float rho = s2 * ir;
  rho = fmaf(0.5f * ir, fmaf(-rho, rho, s2), rho);            // Newton step on sqrt
  const float q = fminf(delta * fminf(ir, inv_eps), 1.0f);    // min(delta / max(rho, eps), 1)
  const float iq = fast_rsqrt(fmaxf(q, 1e-36f));
  float gam = q * iq;
  gam = fmaf(0.5f * iq, fmaf(-gam, gam, q), gam);
  float s0 = p.wu * gam, s1 = p.wv * gam;
  if (clip) {
    const bool zc = (z == z_min);
    bool k0 = zc, k1 = zc;
    if (BOUNDS) {
      k0 = k0 || (px == bd.lbx) || (px == bd.ubx);
      k1 = k1 || (py == bd.lby) || (py == bd.uby);
    }
    s0 = k0 ? 0.f : s0;
    s1 = k1 ? 0.f : s1;
  }
  const float e0 = rx * gam, e1 = ry * gam;
#pragma unroll
  for (int i = 0; i < DOF; ++i) {
    J0[i] *= s0;
    J1[i] *= s1;
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
  for (int i = 0; i < DOF; ++i) acc[NormalEq<DOF>::NH + i] = fmaf(J0[i], e0, fmaf(J1[i], e1, acc[NormalEq<DOF>::NH + i]));
  acc[NormalEq<DOF>::NV - 1] += huber_exact(rho, delta);
}

// workgroup shape chosen by the host for kernels that keep an object's points in registers
struct Shape {
  int waves;   // waves per object (1..16)
  int ppl;     // points per lane (1,2,4,8);  64 * waves * ppl >= N
};

}  // namespace pnp
