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
// PPL >= 1: the object's points are loaded up front into registers (all loads of a lane in flight at once -- this is
// what lets one sweep approach the HBM rate: the loop form below issues load -> 230-instruction body -> load and was
// latency-bound at 2.3 TB/s at C2), 64 * waves * PPL >= N.  PPL == 0: streaming loop for N beyond the resident limit.
template <int DOF, bool BOUNDS, int NV>
__device__ __forceinline__ void store_normal_equations(const float (&acc)[NV], int b, float* __restrict__ jtj,
                                                       float* __restrict__ jtr, float* __restrict__ cost) {
  constexpr int NH = NormalEq<DOF>::NH;
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

template <int DOF, int PPL, bool BOUNDS, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void normal_equations_kernel(Problem p, const float* __restrict__ pose, int clip,
                                                                      float* __restrict__ jtj, float* __restrict__ jtr,
                                                                      float* __restrict__ cost) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NV = NormalEq<DOF>::NV;
  PNP_DYN_SMEM(float, scratch);      // waves * kSumTStride<NV> (transposed reduction) or NV * 16 (DPP fallback)
  const int b = object_of_block(p.B);
  if (b >= p.B) return;
  float K[9], R[9], ps[PL], delta;
  Bounds bd;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  if (PPL > 0) {
    Point pts[PPL > 0 ? PPL : 1];
#pragma unroll
    // non-temporal: the sweep reads every correspondence exactly once (IC-cold at C2: 17.0-17.8 -> 16.0-16.2 us, 0.43 -> 0.46 of
    // the HBM peak; no change at the C5 size, where nothing fits a cache anyway: profiles/r03_tune_ne_nt_loads.txt)
    for (int k = 0; k < PPL; ++k) pts[k] = load_point_streamed(p, b, (int)threadIdx.x + k * (int)blockDim.x);
    load_camera<BOUNDS>(p, b, K, bd, delta);
#pragma unroll
    for (int i = 0; i < PL; ++i) ps[i] = pose[(size_t)b * PL + i];
    pose_to_rot<DOF>(ps, R);
    // wave-uniform operands in VGPRs (an SGPR source operand halves the VALU issue rate on gfx950)
#pragma unroll
    for (int i = 0; i < 9; ++i) { K[i] = to_vgpr(K[i]); R[i] = to_vgpr(R[i]); }
    float t[3] = {to_vgpr(ps[0]), to_vgpr(ps[1]), to_vgpr(ps[2])};
    const float zm = to_vgpr(p.z_min), dl = to_vgpr(delta);
    const float ie = to_vgpr(p.inv_huber_eps);
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      mask_point_beyond(pts[k], (int)threadIdx.x + k * (int)blockDim.x, p.N);
      point_normal_eq<DOF, BOUNDS>(pts[k], K, R, t, zm, dl, ie, bd, clip != 0, acc);
      sched_fence();      // point k's arithmetic under the loads of points k+1.. (wave_ops.h)
    }
  } else {
    load_camera<BOUNDS>(p, b, K, bd, delta);
#pragma unroll
    for (int i = 0; i < PL; ++i) ps[i] = pose[(size_t)b * PL + i];
    pose_to_rot<DOF>(ps, R);
    for (int n = (int)threadIdx.x; n < p.N; n += (int)blockDim.x) {
      const Point q = load_point(p, b, n);
      point_normal_eq<DOF, BOUNDS>(q, K, R, ps, p.z_min, delta, p.inv_huber_eps, bd, clip != 0, acc);
    }
  }
  if (MAXW <= 4) block_sum_t<NV>(acc, scratch); else block_sum<NV>(acc, scratch);
  if (threadIdx.x == 0) store_normal_equations<DOF, BOUNDS, NV>(acc, b, jtj, jtr, cost);
}

// ----------------------------------------------------------------------------------------------------------
// Huber cost of ONE pose over the register-resident points of a lane (this lane's part of the object's sum)
template <int DOF, int PPL, bool BOUNDS>
__device__ __forceinline__ float resident_cost(const Point (&pts)[PPL], const float (&K)[9], const Bounds& bd, float delta,
                                               float z_min, const float (&ps)[PoseLen<DOF>::value]) {
  float R[9], KR[9], Kt[3];
  pose_to_rot<DOF>(ps, R);
  compose_kr_kt(K, R, ps, KR, Kt);
  float c = 0.f;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    // exact Huber form on the IEEE path (matches cost_fun.py:8-12 bit-for-bit per point)
    const Point& q = pts[k];
    // the fma chains are spelled out: left to -ffp-contract=fast, `a x + b y + c z + t` came out as fma(c, z, fma(a, x, b y)) + t
    // in the kernel whose loop over poses hoists the pose-independent product b y (4-DoF: KR[1] = K[1]) and as
    // fma(c, z, fma(b, y, a x)) + t in the one that evaluates a single pose -- the same cost, one ulp apart
    const float hx = add_unfused(fmaf(KR[2], q.Z, fmaf(KR[1], q.Y, mul_unfused(KR[0], q.X))), Kt[0]);
    const float hy = add_unfused(fmaf(KR[5], q.Z, fmaf(KR[4], q.Y, mul_unfused(KR[3], q.X))), Kt[1]);
    const float hz = add_unfused(fmaf(KR[8], q.Z, fmaf(KR[7], q.Y, mul_unfused(KR[6], q.X))), Kt[2]);
    const float z = fmaxf(hz, z_min);
    float px = hx / z, py = hy / z;
    if (BOUNDS) {
      px = clamp_lu(px, bd.lbx, bd.ubx);
      py = clamp_lu(py, bd.lby, bd.uby);
    }
    const float rx = (px - q.u) * q.wu, ry = (py - q.v) * q.wv;
    c += huber_exact(sqrtf(fmaf(rx, rx, mul_unfused(ry, ry))), delta);
  }
  return c;
}

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
        float ps[PL];
#pragma unroll
        for (int i = 0; i < PL; ++i) ps[i] = poses[((size_t)j * p.B + b) * PL + i];
        c[jj] = resident_cost<DOF, PPL, BOUNDS>(pts, K, bd, delta, p.z_min, ps);
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
// Gradients of  sum_j a_j cost(pose_j)  w.r.t. the camera intrinsics and (for one of the poses) w.r.t. the pose -- what the
// reference's autograd records for cost_init = evaluate_pnp(pose=pose_init) and for the AMIS log-weights w.r.t.
// camera.cam_mats and pose_init (epropnp/epropnp.py:121-124,139-169; common.py:90-99; camera.py:21-30,81-93: project_b,
// depth clamp, projection clamp with torch.clamp's masks; cost_fun.py:8-12,45-61: Huber).  With h = K y, y = R X + t and
// g_h = d cost / d h per point-pose:
//     d/dK = sum_j a_j sum_n g_h y^T                                   -> out_gk (B,3,3)
//     M    = a_m sum_n g_h (X,Y,Z,1)^T  for the pose m = m_pose        -> out_m  (B,3,4)
// from which d/dt_m = K^T M[:,3] and d/dR_m = K^T M[:,:3] (host side, epropnp/functional.py:cost_pose_grad).  A rarely
// taken path (nothing in the reference's training loops differentiates w.r.t. these): a plain VALU sweep, lane = point,
// IEEE divide / sqrt as on the cost_init path; poses with zero weight are skipped.
template <int DOF, bool BOUNDS>
__global__ __launch_bounds__(256) void cost_pose_cam_grad_kernel(Problem p, const float* __restrict__ poses,
                                                                 const float* __restrict__ weights, int P, int m_pose,
                                                                 float* __restrict__ out_m, float* __restrict__ out_gk) {
  constexpr int PL = PoseLen<DOF>::value;
  __shared__ float scratch[21 * 4];
  const int b = object_of_block(p.B);
  if (b >= p.B) return;
  float K[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, K, bd, delta);
  float acc[21];          // [0..11] M, [12..20] d/dK
#pragma unroll
  for (int i = 0; i < 21; ++i) acc[i] = 0.f;
  for (int j = 0; j < P; ++j) {
    const float a = (weights != nullptr) ? weights[(size_t)j * p.B + b] : 1.0f;
    if (a == 0.f) continue;                                  // wave-uniform
    float R[9], ps[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) ps[i] = poses[((size_t)j * p.B + b) * PL + i];
    pose_to_rot<DOF>(ps, R);
    float KR[9], Kt[3];
    compose_kr_kt(K, R, ps, KR, Kt);
    const bool want_m = (j == m_pose);
    for (int n = (int)threadIdx.x; n < p.N; n += (int)blockDim.x) {
      const Point q = load_point(p, b, n);
      const float hx = KR[0] * q.X + KR[1] * q.Y + KR[2] * q.Z + Kt[0];
      const float hy = KR[3] * q.X + KR[4] * q.Y + KR[5] * q.Z + Kt[1];
      const float hz = KR[6] * q.X + KR[7] * q.Y + KR[8] * q.Z + Kt[2];
      const bool front = hz >= p.z_min;                    // torch.clamp(min=) passes the gradient where x >= min
      const float z = front ? hz : p.z_min;
      const float ppx = hx / z, ppy = hy / z;
      float px = ppx, py = ppy;
      if (BOUNDS) {
        px = clamp_lu(px, bd.lbx, bd.ubx);
        py = clamp_lu(py, bd.lby, bd.uby);
      }
      const float rx = (px - q.u) * q.wu, ry = (py - q.v) * q.wv;
      const float rho = sqrtf(rx * rx + ry * ry);
      const float coef = a * ((rho <= delta) ? 1.0f : delta / rho);       // a * d huber / d r = a r min(1, delta / rho)
      float gpx = coef * rx * q.wu, gpy = coef * ry * q.wv;
      if (BOUNDS) {                                                       // the clamp passes no gradient where it is active
        gpx = (ppx < bd.lbx || ppx > bd.ubx) ? 0.f : gpx;
        gpy = (ppy < bd.lby || ppy > bd.uby) ? 0.f : gpy;
      }
      const float ghx = gpx / z, ghy = gpy / z;
      const float ghz = front ? -(ghx * ppx + ghy * ppy) : 0.f;
      if (want_m) {
        acc[0] = fmaf(ghx, q.X, acc[0]); acc[1] = fmaf(ghx, q.Y, acc[1]); acc[2] = fmaf(ghx, q.Z, acc[2]); acc[3] += ghx;
        acc[4] = fmaf(ghy, q.X, acc[4]); acc[5] = fmaf(ghy, q.Y, acc[5]); acc[6] = fmaf(ghy, q.Z, acc[6]); acc[7] += ghy;
        acc[8] = fmaf(ghz, q.X, acc[8]); acc[9] = fmaf(ghz, q.Y, acc[9]); acc[10] = fmaf(ghz, q.Z, acc[10]); acc[11] += ghz;
      }
      const float y0 = R[0] * q.X + R[1] * q.Y + R[2] * q.Z + ps[0];
      const float y1 = R[3] * q.X + R[4] * q.Y + R[5] * q.Z + ps[1];
      const float y2 = R[6] * q.X + R[7] * q.Y + R[8] * q.Z + ps[2];
      acc[12] = fmaf(ghx, y0, acc[12]); acc[13] = fmaf(ghx, y1, acc[13]); acc[14] = fmaf(ghx, y2, acc[14]);
      acc[15] = fmaf(ghy, y0, acc[15]); acc[16] = fmaf(ghy, y1, acc[16]); acc[17] = fmaf(ghy, y2, acc[17]);
      acc[18] = fmaf(ghz, y0, acc[18]); acc[19] = fmaf(ghz, y1, acc[19]); acc[20] = fmaf(ghz, y2, acc[20]);
    }
  }
  block_sum<21>(acc, scratch);
  if (threadIdx.x < 21) {
    float v = acc[0];
#pragma unroll
    for (int i = 1; i < 21; ++i) v = ((int)threadIdx.x == i) ? acc[i] : v;
    if (threadIdx.x < 12) {
      if (out_m != nullptr) out_m[(size_t)b * 12 + threadIdx.x] = v;
    } else if (out_gk != nullptr) {
      out_gk[(size_t)b * 9 + (threadIdx.x - 12)] = v;
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// Glue of the training step that the reference leaves to ~25 small ATen launches: adaptive Huber threshold and the
// Monte-Carlo pose loss.  Both are single HBM-bound passes.
__global__ __launch_bounds__(256) void adaptive_delta_kernel(const float* __restrict__ x2d, const float* __restrict__ w2d,
                                                              int B, int N, float rel, float* __restrict__ delta,
                                                              float* __restrict__ stats) {
  __shared__ float red[5 * 16];
  const int b = object_of_block(B);
  if (b >= B) return;
  const float2* x = reinterpret_cast<const float2*>(x2d) + (size_t)b * N;
  const float2* w = reinterpret_cast<const float2*>(w2d) + (size_t)b * N;
  // shifted one-pass moments (pivot = first point) -> no E[x^2] - E[x]^2 cancellation at pixel magnitudes
  const float2 piv = x[0];
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int n = (int)threadIdx.x; n < N; n += (int)blockDim.x) {
    const float2 xi = x[n], wi = w[n];
    const float dx = xi.x - piv.x, dy = xi.y - piv.y;
    v[0] += wi.x + wi.y;
    v[1] += dx; v[2] += dy;
    v[3] += dx * dx; v[4] += dy * dy;
  }
  block_sum<5>(v, red);
  if (threadIdx.x == 0) {
    const float n = (float)N;
    const float var = ((v[3] - v[1] * v[1] / n) + (v[4] - v[2] * v[2] / n)) / (n - 1.0f);
    const float sd = sqrtf(fmaxf(var, 0.f));
    const float mw = v[0] / (2.0f * n);
    delta[b] = mw * sd * rel;
    stats[(size_t)b * 4 + 0] = mw;
    stats[(size_t)b * 4 + 1] = sd;
    stats[(size_t)b * 4 + 2] = piv.x + v[1] / n;
    stats[(size_t)b * 4 + 3] = piv.y + v[2] / n;
  }
}

// (S,B) log-weights: a 512-thread block owns 16 adjacent objects (columns: 64-byte row segments) and splits the S rows
// over 32 row groups; each thread keeps an online (max, sum) pair over batches of 8 independent loads, the 32 partials
// of a column are merged through LDS.
constexpr int kLossCols = 16, kLossRows = 32;
__global__ __launch_bounds__(512) void mc_loss_forward_kernel(const float* __restrict__ logw, const float* __restrict__ ct,
                                                               int S, int B, float* __restrict__ loss,
                                                               float* __restrict__ lse) {
  __shared__ float smax[kLossRows][kLossCols + 1], ssum[kLossRows][kLossCols + 1];
  const int c = (int)(threadIdx.x % kLossCols), rg = (int)(threadIdx.x / kLossCols);
  const int b = (int)blockIdx.x * kLossCols + c;
  float m = -INFINITY, acc = 0.f;
  bool nan = false;
  if (b < B) {
    for (int j0 = rg; j0 < S; j0 += kLossRows * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = j0 + k * kLossRows;
        v[k] = (j < S) ? logw[(size_t)j * B + b] : -INFINITY;
      }
      float cm = -INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        nan = nan || (v[k] != v[k]);
        cm = (v[k] > cm) ? v[k] : cm;
      }
      if (cm > m) {
        acc *= expf(m - cm);                  // exp(-inf - x) = 0 on the first batch
        m = cm;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        acc += (v[k] == -INFINITY || v[k] != v[k]) ? 0.f : ((v[k] == m) ? 1.0f : expf(v[k] - m));
    }
  }
  smax[rg][c] = nan ? NAN : m;
  ssum[rg][c] = acc;
  __syncthreads();
  if (rg == 0 && b < B) {
    float M = -INFINITY;
    bool bad = false;
    for (int k = 0; k < kLossRows; ++k) {
      const float mk = smax[k][c];
      bad = bad || (mk != mk);
      M = fmaxf(M, mk);
    }
    float tot = 0.f;
    for (int k = 0; k < kLossRows; ++k) {
      const float mk = smax[k][c];
      if (mk == mk && mk > -INFINITY) tot += ssum[k][c] * ((mk == M) ? 1.0f : expf(mk - M));
    }
    float l = (M == -INFINITY || M == INFINITY) ? M : M + logf(tot);
    if (bad) l = NAN;
    const float v = l + (ct ? ct[b] : 0.f);
    lse[b] = (v != v) ? NAN : l;          // NaN marks "loss zeroed": the backward passes no gradient there
    loss[b] = (v != v) ? 0.f : v;
  }
}

// g == nullptr: the reduced loss (mc_loss_reduce_kernel) -- every object's upstream gradient is the scalar
// grad_out[0] * coef[0] (* weight[b]), read from device memory so that the node stays capturable.
__global__ __launch_bounds__(256) void mc_loss_backward_kernel(const float* __restrict__ logw, const float* __restrict__ lse,
                                                                const float* __restrict__ g, const float* __restrict__ gout,
                                                                const float* __restrict__ coef, const float* __restrict__ weight,
                                                                int S, int B, float* __restrict__ glogw,
                                                                float* __restrict__ gct) {
  const size_t total = (size_t)S * B;
  const float gs = (g == nullptr) ? gout[0] * coef[0] : 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % (size_t)B);
    const float l = lse[b];
    const float gb = (g != nullptr) ? g[b] : (weight != nullptr ? gs * weight[b] : gs);
    glogw[i] = (l != l) ? 0.f : gb * expf(logw[i] - l);
    if (gct != nullptr && i < (size_t)B) gct[b] = (l != l) ? 0.f : gb;      // the first B threads cover b = 0..B-1
  }
}

// The scalar both reference loss modules return, from the per-object losses, in ONE single-workgroup launch instead of the
// ~10 elementwise / reduce launches of its PyTorch statement (EPro-PnP-Det epropnp_det/models/losses/monte_carlo_pose_loss.py:
// 41-66 with mmdet's weight_reduce_loss; EPro-PnP-6DoF lib/models/monte_carlo_pose_loss.py:20-35):
//     norm_factor <- (1 - momentum) norm_factor + momentum * norm_factor_in          (training: nf_in != nullptr)
//     out[0] = (sum_b weight[b] loss[b]) * scale / norm_factor,   out[1] = scale / norm_factor   (kept for the backward)
// scale = loss_weight / B (mean) | loss_weight (sum) | loss_weight / avg_factor.  The sum runs in a fixed order (thread t takes
// b = t, t + T, ...; then the block tree): bit-reproducible.  The running estimate is written with separately rounded
// products, as the reference's mul_ / add_ pair does.
__global__ __launch_bounds__(1024) void mc_loss_reduce_kernel(const float* __restrict__ loss, const float* __restrict__ weight,
                                                               int B, float scale, float one_minus_m, float m,
                                                               const float* __restrict__ nf_in, int nf_count, long long nf_stride,
                                                               float* __restrict__ nf, float* __restrict__ out) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int b = (int)threadIdx.x; b < B; b += (int)blockDim.x) acc += (weight != nullptr) ? loss[b] * weight[b] : loss[b];
  float v[1] = {acc};
  block_sum<1>(v, red);
  if (threadIdx.x == 0) {
    float n = (nf != nullptr) ? nf[0] : 1.0f;
    if (nf != nullptr && nf_in != nullptr) {
      float in = nf_in[0];
      if (nf_count > 1) {       // the world mean of the ranks' inputs, read out of the all-gather's receive buffer in rank order
        for (int r = 1; r < nf_count; ++r) in += nf_in[(size_t)r * (size_t)nf_stride];
        in /= (float)nf_count;
      }
      n = add_unfused(mul_unfused(n, one_minus_m), mul_unfused(m, in));       // the reference's two roundings
      nf[0] = n;
    }
    const float c = scale / n;
    out[0] = v[0] * c;
    out[1] = c;
  }
}

// One wave per (proposal, object) row.  Keys live in LDS; each of the n_pts rounds is a wave-wide argmin.
// N <= 512 uses the packed keys of pnp_math.h (race_key): one integer min per round, the same stream and the same
// winners as the fused initialiser (rslm_kernel.hip).  Larger N keeps float keys with an explicit lowest-index tie-break.
__global__ __launch_bounds__(64) void rslm_draw_kernel(const float* __restrict__ w2d, int B, int N, int P, int n_pts,
                                                        unsigned long long seed, unsigned long long offset,
                                                        long long* __restrict__ inds) {
  PNP_DYN_SMEM(float, key);
  unsigned* ukey = reinterpret_cast<unsigned*>(key);
  const int row = (int)blockIdx.x;          // row = proposal * B + object
  const int b = row % B;
  const int lane = lane_id();
  const bool packed = N <= (int)(kRaceIdxMask + 1);
  const float2* w = reinterpret_cast<const float2*>(w2d) + (size_t)b * N;
  for (int n4 = lane; 4 * n4 < N; n4 += 64) {      // one Philox block = the uniforms of points 4 n4 .. 4 n4 + 3
    const Philox4 r = philox4x32_10((uint32_t)row, (uint32_t)n4, (uint32_t)offset, (uint32_t)(offset >> 32),
                                    (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5bd1e995u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 4 * n4 + q;
      if (n < N) {
        const float2 wi = w[n];
        const float wm = 0.5f * (wi.x + wi.y);
        if (packed) ukey[n] = race_key(r.v[q], race_inv_weight(wm), n);
        else key[n] = (wm > 0.f) ? fabsf(logf(u01(r.v[q]))) / wm : INFINITY;
      }
    }
  }
  wave_lds_fence();
  for (int k = 0; k < n_pts; ++k) {
    int wi;
    if (packed) {
      unsigned best = 0xffffffffu;
      for (int n = lane; n < N; n += 64) best = min(best, ukey[n]);
      const unsigned m = wave_min_u32(best);
      const int slot = (int)(m & kRaceIdxMask);
      wi = (m < kRaceInf) ? slot : (k % N);   // fewer positive weights than n_pts: fall back deterministically
      if (m != 0xffffffffu && lane == (slot & 63)) ukey[slot] = 0xffffffffu;
    } else {
      float best = INFINITY, best_n = 1e9f;
      for (int n = lane; n < N; n += 64) {
        const float v = key[n];
        if (v < best) { best = v; best_n = (float)n; }
      }
      const float m = -wave_max(-best);
      const float cand = (best == m) ? best_n : 1e9f;
      const float win = -wave_max(-cand);       // lowest index among ties
      wi = (win < 1e8f) ? (int)win : (k % N);
      if (lane == (wi & 63)) key[wi] = INFINITY;
    }
    if (lane == 0) inds[(size_t)row * n_pts + k] = (long long)wi;
    wave_lds_fence();
  }
}

// pnp_normalize / pnp_denormalize (epropnp/common.py:103-136) as two launches instead of ~10 ATen launches each.
// center: offset[b] = mean_n x3d[b,n,:];  out[b,n,:] = x3d[b,n,:] - offset[b]
// DOF != 0: the object's pose_init is moved into the centred frame by the same launch (shift_poses_kernel's arithmetic with
// sign = +1, by thread 0) -- pnp_normalize's two steps in one launch of the one-call forward.
// (block size: center_threads() -- the thread count fixes the order of the mean's sum, and center_cost_kernel below must
// reproduce it)
template <int DOF>
__global__ __launch_bounds__(1024) void center_points_kernel(const float* __restrict__ x3d, int B, int N,
                                                              float* __restrict__ offset, float* __restrict__ out,
                                                              const float* __restrict__ pose, float* __restrict__ pose_out) {
  __shared__ float scratch[3 * 16];
  const int b = object_of_block(B);
  if (b >= B) return;
  const float* src = x3d + (size_t)b * N * 3;
  float s[3] = {0.f, 0.f, 0.f};
  for (int n = (int)threadIdx.x; n < N; n += (int)blockDim.x) {
    s[0] += src[3 * n]; s[1] += src[3 * n + 1]; s[2] += src[3 * n + 2];
  }
  block_sum<3>(s, scratch);
  const float m0 = s[0] / (float)N, m1 = s[1] / (float)N, m2 = s[2] / (float)N;
  float* dst = out + (size_t)b * N * 3;
  for (int n = (int)threadIdx.x; n < N; n += (int)blockDim.x) {
    dst[3 * n] = src[3 * n] - m0; dst[3 * n + 1] = src[3 * n + 1] - m1; dst[3 * n + 2] = src[3 * n + 2] - m2;
  }
  if (threadIdx.x == 0) {
    offset[(size_t)b * 3] = m0; offset[(size_t)b * 3 + 1] = m1; offset[(size_t)b * 3 + 2] = m2;
    if (DOF != 0) {
      constexpr int PL = PoseLen<DOF == 0 ? 6 : DOF>::value;
      float ps[PL], R[9];
#pragma unroll
      for (int k = 0; k < PL; ++k) ps[k] = pose[(size_t)b * PL + k];
      pose_to_rot<DOF == 0 ? 6 : DOF>(ps, R);
      shift_translation(ps, R, m0, m1, m2, 1.0f);
#pragma unroll
      for (int k = 0; k < PL; ++k) pose_out[(size_t)b * PL + k] = ps[k];
    }
  }
}

// pnp_normalize of the points and of pose_init (center_points_kernel<DOF>) AND the cost of pose_init in the centred frame
// (evaluate_cost_kernel with one pose) in ONE launch: the first two launches of the one-call forward with normalize=True at the
// launch-bound shapes (EPro-PnP-Det: 600 x 128).  The points are loaded once into registers, summed in center_points_kernel's
// order (same block size: center_threads), centred in place and written out; the cost is resident_cost on those registers and
// the same block_sum<4> -- offset, centred points, pose and cost are bit-identical to the two separate launches.
template <int DOF, int PPL, bool BOUNDS, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void center_cost_kernel(Problem p, const float* __restrict__ pose,
                                                                  float* __restrict__ offset, float* __restrict__ x3d_out,
                                                                  float* __restrict__ pose_out, float* __restrict__ cost) {
  constexpr int PL = PoseLen<DOF>::value;
  __shared__ float scratch[4 * 16];
  const int b = object_of_block(p.B);
  if (b >= p.B) return;
  float K[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, K, bd, delta);
  Point pts[PPL];
  float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int n = (int)threadIdx.x + k * (int)blockDim.x;
    pts[k] = load_point(p, b, n);
    if (n < p.N) { s[0] += pts[k].X; s[1] += pts[k].Y; s[2] += pts[k].Z; }
  }
  block_sum<3>(s, scratch);
  const float m0 = s[0] / (float)p.N, m1 = s[1] / (float)p.N, m2 = s[2] / (float)p.N;
  float* dst = x3d_out + (size_t)b * p.N * 3;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int n = (int)threadIdx.x + k * (int)blockDim.x;
    if (n < p.N) {      // (padding keeps its zero coordinates and zero weights, as evaluate_cost_kernel would load it)
      pts[k].X -= m0; pts[k].Y -= m1; pts[k].Z -= m2;
      dst[3 * n] = pts[k].X; dst[3 * n + 1] = pts[k].Y; dst[3 * n + 2] = pts[k].Z;
    }
  }
  float ps[PL], R[9];
#pragma unroll
  for (int k = 0; k < PL; ++k) ps[k] = pose[(size_t)b * PL + k];
  pose_to_rot<DOF>(ps, R);
  shift_translation(ps, R, m0, m1, m2, 1.0f);
  if (threadIdx.x == 0) {
    offset[(size_t)b * 3] = m0; offset[(size_t)b * 3 + 1] = m1; offset[(size_t)b * 3 + 2] = m2;
#pragma unroll
    for (int k = 0; k < PL; ++k) pose_out[(size_t)b * PL + k] = ps[k];
  }
  float c[4] = {resident_cost<DOF, PPL, BOUNDS>(pts, K, bd, delta, p.z_min, ps), 0.f, 0.f, 0.f};
  block_sum<4>(c, scratch);
  if (threadIdx.x == 0) cost[b] = c[0];
}

// shift: out[j,b] = pose[j,b] with translation += sign * R(pose[j,b]) offset[b]
template <int DOF>
__global__ __launch_bounds__(256) void shift_poses_kernel(const float* __restrict__ pose, const float* __restrict__ offset,
                                                           int P, int B, float sign, float* __restrict__ out) {
  constexpr int PL = PoseLen<DOF>::value;
  const size_t total = (size_t)P * B;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % (size_t)B);
    float ps[PL], R[9];
#pragma unroll
    for (int k = 0; k < PL; ++k) ps[k] = pose[i * PL + k];
    pose_to_rot<DOF>(ps, R);
    const float ox = offset[(size_t)b * 3], oy = offset[(size_t)b * 3 + 1], oz = offset[(size_t)b * 3 + 2];
    shift_translation(ps, R, ox, oy, oz, sign);
#pragma unroll
    for (int k = 0; k < PL; ++k) out[i * PL + k] = ps[k];
  }
}

// the two shifts of pnp_denormalize (pose_opt and the S x B samples, common.py:127-136) in ONE launch: at the detection shape
// each is a ~5 us launch around ~1 us of work
template <int DOF>
__global__ __launch_bounds__(256) void shift_poses_pair_kernel(const float* __restrict__ pose_a, float* __restrict__ out_a, int Pa,
                                                                const float* __restrict__ pose_b, float* __restrict__ out_b, int Pb,
                                                                const float* __restrict__ offset, int B, float sign) {
  constexpr int PL = PoseLen<DOF>::value;
  const size_t na = (size_t)Pa * B, total = na + (size_t)Pb * B;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const bool first = g < na;
    const size_t i = first ? g : g - na;
    const float* __restrict__ pose = first ? pose_a : pose_b;
    float* __restrict__ out = first ? out_a : out_b;
    const int b = (int)(i % (size_t)B);
    float ps[PL], R[9];
#pragma unroll
    for (int k = 0; k < PL; ++k) ps[k] = pose[i * PL + k];
    pose_to_rot<DOF>(ps, R);
    const float ox = offset[(size_t)b * 3], oy = offset[(size_t)b * 3 + 1], oz = offset[(size_t)b * 3 + 2];
    shift_translation(ps, R, ox, oy, oz, sign);
#pragma unroll
    for (int k = 0; k < PL; ++k) out[i * PL + k] = ps[k];
  }
}

// backward of shift_poses w.r.t. the pose (the offset is a detached constant, as in pnp_normalize):
//   out_t = t + sign * R(rot) o   =>   g_t passes through;  g_rot += sign * d(R o)/d(rot)^T g_t
//   6-DoF, R o = (w^2 - v.v) o + 2 v (v.o) + 2 w (v x o)  (common.py:21-42, q not normalised):
//     d/dw = 2 w o + 2 v x o,   grad_v = -2 (g.o) v + 2 (v.o) g + 2 (g.v) o + 2 w (o x g)
//   4-DoF, R = Ry(yaw):  d(R o)/dyaw = (-s ox + c oz, 0, -c ox - s oz)
template <int DOF>
__global__ __launch_bounds__(256) void shift_poses_backward_kernel(const float* __restrict__ pose, const float* __restrict__ offset,
                                                                    const float* __restrict__ gout, int P, int B, float sign,
                                                                    float* __restrict__ gpose) {
  constexpr int PL = PoseLen<DOF>::value;
  const size_t total = (size_t)P * B;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % (size_t)B);
    float ps[PL], go[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k) { ps[k] = pose[i * PL + k]; go[k] = gout[i * PL + k]; }
    const float ox = offset[(size_t)b * 3], oy = offset[(size_t)b * 3 + 1], oz = offset[(size_t)b * 3 + 2];
    const float gx = sign * go[0], gy = sign * go[1], gz = sign * go[2];
    float gp[PL];
    gp[0] = go[0]; gp[1] = go[1]; gp[2] = go[2];
    if (DOF == 4) {
      const float c = cosf(ps[3]), s = sinf(ps[3]);
      gp[3] = go[3] + gx * (-s * ox + c * oz) + gz * (-c * ox - s * oz);
    } else {
      const float w = ps[3], vx = ps[4], vy = ps[5], vz = ps[6];
      const float cx = vy * oz - vz * oy, cy = vz * ox - vx * oz, cz = vx * oy - vy * ox;      // v x o
      const float g_o = gx * ox + gy * oy + gz * oz, v_o = vx * ox + vy * oy + vz * oz, g_v = gx * vx + gy * vy + gz * vz;
      const float ogx = oy * gz - oz * gy, ogy = oz * gx - ox * gz, ogz = ox * gy - oy * gx;   // o x g
      gp[3] = go[3] + 2.f * (w * g_o + (gx * cx + gy * cy + gz * cz));
      gp[4] = go[4] + 2.f * (-g_o * vx + v_o * gx + g_v * ox + w * ogx);
      gp[5] = go[5] + 2.f * (-g_o * vy + v_o * gy + g_v * oy + w * ogy);
      gp[6] = go[6] + 2.f * (-g_o * vz + v_o * gz + g_v * oz + w * ogz);
    }
#pragma unroll
    for (int k = 0; k < PL; ++k) gpose[i * PL + k] = gp[k];
  }
}

int launch_shift_poses_backward(const float* pose, const float* offset, const float* gout, int P, int B, int dof, float sign,
                                float* gpose, hipStream_t st) {
  if (B <= 0 || P <= 0) return EPROPNP_OK;
  if (!pose || !offset || !gout || !gpose || (dof != 4 && dof != 6))
    return fail(EPROPNP_EINVAL, "shift_poses_backward: bad argument");
  const size_t total = (size_t)P * B;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dof == 6) PNP_LAUNCH(shift_poses_backward_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, st, pose, offset, gout, P, B, sign, gpose);
  else PNP_LAUNCH(shift_poses_backward_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, pose, offset, gout, P, B, sign, gpose);
  return check_launch("shift_poses_backward_kernel");
}

// threads per object of the centring kernels: the block size of the register-resident sweeps (choose_shape) wherever those
// can hold the object -- the fused centre + cost launch sums the mean in that order -- and 256 beyond
static int center_threads(int B, int N) { return N <= kMaxResidentPoints ? 64 * choose_shape(B, N).waves : 256; }

int launch_center_points(const float* x3d, int B, int N, float* offset, float* out, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!x3d || !offset || !out || N < 1) return fail(EPROPNP_EINVAL, "center_points: bad argument");
  PNP_LAUNCH(center_points_kernel<0>, dim3(padded_object_grid(B)), dim3(center_threads(B, N)), 0, st, x3d, B, N, offset, out,
             (const float*)nullptr, (float*)nullptr);
  return check_launch("center_points_kernel");
}

int launch_center_points_shift(const float* x3d, int B, int N, float* offset, float* out, const float* pose, float* pose_out,
                               int dof, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!x3d || !offset || !out || !pose || !pose_out || N < 1 || (dof != 4 && dof != 6))
    return fail(EPROPNP_EINVAL, "center_points_shift: bad argument");
  const int threads = center_threads(B, N);
  if (dof == 6)
    PNP_LAUNCH(center_points_kernel<6>, dim3(padded_object_grid(B)), dim3(threads), 0, st, x3d, B, N, offset, out, pose, pose_out);
  else
    PNP_LAUNCH(center_points_kernel<4>, dim3(padded_object_grid(B)), dim3(threads), 0, st, x3d, B, N, offset, out, pose, pose_out);
  return check_launch("center_points_kernel (+ pose_init)");
}

// prob: the problem on the caller's (uncentred) points.  -> offset (B,3), x3d_centered (B,N,3), pose_n (B,PL), cost (B,) of pose_n
// on the centred points.  num_pts <= the register-resident limit (as evaluate_cost).
int launch_center_cost(const epropnp_problem* prob, const float* pose, float* offset, float* x3d_centered, float* pose_n,
                       float* cost, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose || !offset || !x3d_centered || !pose_n || !cost) return fail(EPROPNP_EINVAL, "center_cost: NULL pointer");
  if (prob->num_pts > kMaxResidentPoints)
    return fail(EPROPNP_EINVAL, "center_cost: num_pts %d exceeds the register-resident limit %d", prob->num_pts, kMaxResidentPoints);
  const Problem d = to_device_problem(prob);
  const Shape s = choose_shape(d.B, d.N);
  const dim3 grid(padded_object_grid(d.B)), block(64 * s.waves);
  dispatch_shape(prob->dof, s.ppl, has_bounds(prob), s.waves, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    PNP_LAUNCH((center_cost_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value, decltype(MAXW)::value>),
               grid, block, 0, st, d, pose, offset, x3d_centered, pose_n, cost);
    return 0;
  });
  return check_launch("center_cost_kernel");
}

int launch_shift_poses(const float* pose, const float* offset, int P, int B, int dof, float sign, float* out,
                       hipStream_t st) {
  if (B <= 0 || P <= 0) return EPROPNP_OK;
  if (!pose || !offset || !out || (dof != 4 && dof != 6)) return fail(EPROPNP_EINVAL, "shift_poses: bad argument");
  const size_t total = (size_t)P * B;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dof == 6) PNP_LAUNCH(shift_poses_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, st, pose, offset, P, B, sign, out);
  else PNP_LAUNCH(shift_poses_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, pose, offset, P, B, sign, out);
  return check_launch("shift_poses_kernel");
}

int launch_shift_poses_pair(const float* pose_a, float* out_a, int Pa, const float* pose_b, float* out_b, int Pb,
                            const float* offset, int B, int dof, float sign, hipStream_t st) {
  if (B <= 0 || Pa + Pb <= 0) return EPROPNP_OK;
  if (!pose_a || !out_a || !pose_b || !out_b || !offset) return fail(EPROPNP_EINVAL, "shift_poses_pair: NULL pointer");
  const size_t total = (size_t)(Pa + Pb) * B;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dof == 6) PNP_LAUNCH(shift_poses_pair_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, st, pose_a, out_a, Pa, pose_b, out_b, Pb, offset, B, sign);
  else PNP_LAUNCH(shift_poses_pair_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, pose_a, out_a, Pa, pose_b, out_b, Pb, offset, B, sign);
  return check_launch("shift_poses_pair_kernel");
}

// Correspondence pre-processing of the reference's callers, fused (forward + backward):
//   x3d = noc * dim                                   (EPro-PnP-6DoF/lib/train.py:141, Det deform_pnp_head.py:873)
//   mode 0: w2d = softmax_N(logits) * scale           (Det deform_pnp_head.py:418-423,874)
//   mode 1: w2d = exp(logits - mean_N(logits) - log N) * scale     ("mean-normalised exp", lib/train.py:163-166)
// One workgroup per object; stats[b] = {max or mean (x, y), sum of exponentials (x, y)} is kept for the backward.
//
// Two input layouts.  Point-major: noc (B,N,3), logits (B,N,2).  Dense (lib/train.py:143-162): the network's maps
// noc (B,3,H*W), logits (B,2,H*W) gathered at inds (B,N) -- `x.flatten(2).transpose(-1,-2)[batch_inds, sample_inds]`
// without materialising the transposed maps -- plus the pixel grid x2d = wh_begin + (col, row) * wh_unit of the sampled
// pixels (box = [begin_x, begin_y, unit] per object).
struct PrepLayout {
  const long long* inds;   // (B,N) pixel index row * W + col, or nullptr (point-major)
  int HW, W;
};

// begin + index * unit with TWO roundings, as the reference's separate torch mul and add (HIP's __fmul_rn is a plain `*`
// that -ffp-contract=fast fuses with the add)
__device__ __forceinline__ float mul_then_add(float a, float b, float c) { return add_unfused(c, mul_unfused(a, b)); }

__device__ __forceinline__ float2 prep_ld2(const float* src, int b, int n, int N, const PrepLayout& L) {
  if (L.inds == nullptr) return reinterpret_cast<const float2*>(src)[(size_t)b * N + n];
  const size_t o = (size_t)b * 2 * L.HW + (size_t)L.inds[(size_t)b * N + n];
  return make_float2(src[o], src[o + L.HW]);
}
__device__ __forceinline__ void prep_ld3(const float* src, int b, int n, int N, const PrepLayout& L, float (&v)[3]) {
  if (L.inds == nullptr) {
    const size_t o = ((size_t)b * N + n) * 3;
    v[0] = src[o]; v[1] = src[o + 1]; v[2] = src[o + 2];
  } else {
    const size_t o = (size_t)b * 3 * L.HW + (size_t)L.inds[(size_t)b * N + n];
    v[0] = src[o]; v[1] = src[o + L.HW]; v[2] = src[o + 2 * (size_t)L.HW];
  }
}
// gradients of the dense maps are scattered into zero-filled buffers (atomics: `inds` may repeat a pixel)
__device__ __forceinline__ void prep_st2(float* dst, int b, int n, int N, const PrepLayout& L, float x, float y) {
  if (L.inds == nullptr) {
    reinterpret_cast<float2*>(dst)[(size_t)b * N + n] = make_float2(x, y);
  } else {
    const size_t o = (size_t)b * 2 * L.HW + (size_t)L.inds[(size_t)b * N + n];
    atomicAdd(dst + o, x); atomicAdd(dst + o + L.HW, y);
  }
}
__device__ __forceinline__ void prep_st3(float* dst, int b, int n, int N, const PrepLayout& L, float x, float y, float z) {
  if (L.inds == nullptr) {
    const size_t o = ((size_t)b * N + n) * 3;
    dst[o] = x; dst[o + 1] = y; dst[o + 2] = z;
  } else {
    const size_t o = (size_t)b * 3 * L.HW + (size_t)L.inds[(size_t)b * N + n];
    atomicAdd(dst + o, x); atomicAdd(dst + o + L.HW, y); atomicAdd(dst + o + 2 * (size_t)L.HW, z);
  }
}

__global__ __launch_bounds__(256) void prepare_forward_kernel(const float* __restrict__ noc, const float* __restrict__ dim,
                                                               const float* __restrict__ logits,
                                                               const float* __restrict__ scale, PrepLayout L,
                                                               const float* __restrict__ box, int B, int N, int mode,
                                                               float* __restrict__ x3d, float* __restrict__ x2d,
                                                               float* __restrict__ w2d, float* __restrict__ stats) {
  __shared__ float scratch[4 * 4];
  const int b = object_of_block(B);
  if (b >= B) return;
  const int T = (int)blockDim.x, tid = (int)threadIdx.x;
  float ref[2];
  if (mode == 0) {
    float mx = -INFINITY, my = -INFINITY;
    for (int n = tid; n < N; n += T) {
      const float2 v = prep_ld2(logits, b, n, N, L);
      mx = fmaxf(mx, v.x); my = fmaxf(my, v.y);
    }
    ref[0] = block_max(mx, scratch);
    ref[1] = block_max(my, scratch);
  } else {
    float s[2] = {0.f, 0.f};
    for (int n = tid; n < N; n += T) {
      const float2 v = prep_ld2(logits, b, n, N, L);
      s[0] += v.x; s[1] += v.y;
    }
    block_sum<2>(s, scratch);
    ref[0] = s[0] / (float)N; ref[1] = s[1] / (float)N;
  }
  float se[2] = {0.f, 0.f};
  if (mode == 0) {
    for (int n = tid; n < N; n += T) {
      const float2 v = prep_ld2(logits, b, n, N, L);
      se[0] += expf(v.x - ref[0]); se[1] += expf(v.y - ref[1]);
    }
    block_sum<2>(se, scratch);
  } else {
    se[0] = se[1] = (float)N;                // exp(l - mean - log N) = exp(l - mean) / N
  }
  const float sx = scale ? scale[(size_t)b * 2] : 1.f, sy = scale ? scale[(size_t)b * 2 + 1] : 1.f;
  const float kx = sx / se[0], ky = sy / se[1];
  float2* wo = reinterpret_cast<float2*>(w2d) + (size_t)b * N;
  for (int n = tid; n < N; n += T) {
    const float2 v = prep_ld2(logits, b, n, N, L);
    wo[n] = make_float2(expf(v.x - ref[0]) * kx, expf(v.y - ref[1]) * ky);
  }
  if (x3d != nullptr) {
    const float d0 = dim[(size_t)b * 3], d1 = dim[(size_t)b * 3 + 1], d2 = dim[(size_t)b * 3 + 2];
    float* dst = x3d + (size_t)b * N * 3;
    for (int n = tid; n < N; n += T) {
      float v[3];
      prep_ld3(noc, b, n, N, L, v);
      dst[3 * n] = v[0] * d0; dst[3 * n + 1] = v[1] * d1; dst[3 * n + 2] = v[2] * d2;
    }
  }
  if (x2d != nullptr) {     // pixel grid of the sampled pixels: begin + index * unit, rounded as the reference's mul, add
    const float bx = box[(size_t)b * 3], by = box[(size_t)b * 3 + 1], unit = box[(size_t)b * 3 + 2];
    float2* xo = reinterpret_cast<float2*>(x2d) + (size_t)b * N;
    for (int n = tid; n < N; n += T) {
      const int idx = (int)L.inds[(size_t)b * N + n];          // < H * W: 32-bit divide
      const float col = (float)(idx % L.W), row = (float)(idx / L.W);
      xo[n] = make_float2(mul_then_add(col, unit, bx), mul_then_add(row, unit, by));
    }
  }
  if (tid == 0) {
    stats[(size_t)b * 4] = ref[0]; stats[(size_t)b * 4 + 1] = ref[1];
    stats[(size_t)b * 4 + 2] = se[0]; stats[(size_t)b * 4 + 3] = se[1];
  }
}

// grad_x3d (B,N,3) | NULL, grad_w2d (B,N,2) -> grad_noc, grad_dim (B,3), grad_logits, grad_scale (B,2); grad_noc and
// grad_logits have the layout of the forward's inputs (dense: zero-filled by the launcher, scattered here).
__global__ __launch_bounds__(256) void prepare_backward_kernel(const float* __restrict__ noc, const float* __restrict__ dim,
                                                                const float* __restrict__ logits,
                                                                const float* __restrict__ scale, PrepLayout L,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ gx3d,
                                                                const float* __restrict__ gw2d, int B, int N, int mode,
                                                                float* __restrict__ gnoc, float* __restrict__ gdim,
                                                                float* __restrict__ glogits, float* __restrict__ gscale) {
  __shared__ float scratch[5 * 4];
  const int b = object_of_block(B);
  if (b >= B) return;
  const int T = (int)blockDim.x, tid = (int)threadIdx.x;
  const float2* gw = reinterpret_cast<const float2*>(gw2d) + (size_t)b * N;
  const float r0 = stats[(size_t)b * 4], r1 = stats[(size_t)b * 4 + 1];
  const float ie0 = 1.0f / stats[(size_t)b * 4 + 2], ie1 = 1.0f / stats[(size_t)b * 4 + 3];
  const float sx = scale ? scale[(size_t)b * 2] : 1.f, sy = scale ? scale[(size_t)b * 2 + 1] : 1.f;
  // sums: sum_n g_n p_n per channel (p = normalised exponential without the scale), sum_n gx3d_n * noc_n per axis
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int n = tid; n < N; n += T) {
    const float2 v = prep_ld2(logits, b, n, N, L), g = gw[n];
    s[0] = fmaf(g.x, expf(v.x - r0) * ie0, s[0]);
    s[1] = fmaf(g.y, expf(v.y - r1) * ie1, s[1]);
    if (gx3d != nullptr) {
      const size_t o = ((size_t)b * N + n) * 3;
      float c[3];
      prep_ld3(noc, b, n, N, L, c);
      s[2] = fmaf(gx3d[o], c[0], s[2]); s[3] = fmaf(gx3d[o + 1], c[1], s[3]); s[4] = fmaf(gx3d[o + 2], c[2], s[4]);
    }
  }
  block_sum<5>(s, scratch);
  // softmax:          dL/dl_n = s p_n (g_n - sum_m g_m p_m)
  // mean-normalised:  dL/dl_n = s (g_n p_n - (1/N) sum_m g_m p_m)
  const float invn = 1.0f / (float)N;
  for (int n = tid; n < N; n += T) {
    const float2 v = prep_ld2(logits, b, n, N, L), g = gw[n];
    const float px = expf(v.x - r0) * ie0, py = expf(v.y - r1) * ie1;
    float ox, oy;
    if (mode == 0) {
      ox = sx * px * (g.x - s[0]); oy = sy * py * (g.y - s[1]);
    } else {
      ox = sx * (g.x * px - invn * s[0]); oy = sy * (g.y * py - invn * s[1]);
    }
    prep_st2(glogits, b, n, N, L, ox, oy);
  }
  if (gx3d != nullptr) {
    const float d0 = dim[(size_t)b * 3], d1 = dim[(size_t)b * 3 + 1], d2 = dim[(size_t)b * 3 + 2];
    for (int n = tid; n < N; n += T) {
      const size_t o = ((size_t)b * N + n) * 3;
      prep_st3(gnoc, b, n, N, L, gx3d[o] * d0, gx3d[o + 1] * d1, gx3d[o + 2] * d2);
    }
  }
  if (tid == 0) {
    if (gscale) { gscale[(size_t)b * 2] = s[0]; gscale[(size_t)b * 2 + 1] = s[1]; }
    if (gdim && gx3d != nullptr) { gdim[(size_t)b * 3] = s[2]; gdim[(size_t)b * 3 + 1] = s[3]; gdim[(size_t)b * 3 + 2] = s[4]; }
  }
}

static int prepare_threads(int N) {
  int t = 64;
  while (t < 256 && t * 2 < N) t *= 2;
  return t;
}

int launch_prepare_forward(const float* noc, const float* dim, const float* logits, const float* scale, int B, int N,
                           int mode, float* x3d, float* w2d, float* stats, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!logits || !w2d || !stats || N < 1 || (mode != 0 && mode != 1) || ((x3d != nullptr) && (!noc || !dim)))
    return fail(EPROPNP_EINVAL, "prepare_forward: bad argument");
  const PrepLayout L = {nullptr, 0, 0};
  PNP_LAUNCH(prepare_forward_kernel, dim3(padded_object_grid(B)), dim3(prepare_threads(N)), 0, st, noc, dim, logits, scale, L,
             (const float*)nullptr, B, N, mode, x3d, (float*)nullptr, w2d, stats);
  return check_launch("prepare_forward_kernel");
}

int launch_prepare_backward(const float* noc, const float* dim, const float* logits, const float* scale, const float* stats,
                            const float* gx3d, const float* gw2d, int B, int N, int mode, float* gnoc, float* gdim,
                            float* glogits, float* gscale, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!logits || !stats || !gw2d || !glogits || N < 1 || (mode != 0 && mode != 1) ||
      ((gx3d != nullptr) && (!noc || !dim || !gnoc)))
    return fail(EPROPNP_EINVAL, "prepare_backward: bad argument");
  const PrepLayout L = {nullptr, 0, 0};
  PNP_LAUNCH(prepare_backward_kernel, dim3(padded_object_grid(B)), dim3(prepare_threads(N)), 0, st, noc, dim, logits, scale, L,
             stats, gx3d, gw2d, B, N, mode, gnoc, gdim, glogits, gscale);
  return check_launch("prepare_backward_kernel");
}

int launch_prepare_dense_forward(const float* noc_map, const float* dim, const float* logit_map, const float* scale,
                                 const float* box, const long long* inds, int B, int N, int H, int W, int mode, float* x3d,
                                 float* x2d, float* w2d, float* stats, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!logit_map || !inds || !w2d || !stats || N < 1 || H < 1 || W < 1 || (mode != 0 && mode != 1) ||
      ((x3d != nullptr) && (!noc_map || !dim)) || ((x2d != nullptr) && !box))
    return fail(EPROPNP_EINVAL, "prepare_dense_forward: bad argument");
  const PrepLayout L = {inds, H * W, W};
  PNP_LAUNCH(prepare_forward_kernel, dim3(padded_object_grid(B)), dim3(prepare_threads(N)), 0, st, noc_map, dim, logit_map,
             scale, L, box, B, N, mode, x3d, x2d, w2d, stats);
  return check_launch("prepare_forward_kernel (dense)");
}

int launch_prepare_dense_backward(const float* noc_map, const float* dim, const float* logit_map, const float* scale,
                                  const long long* inds, const float* stats, const float* gx3d, const float* gw2d, int B,
                                  int N, int H, int W, int mode, float* gnoc_map, float* gdim, float* glogit_map,
                                  float* gscale, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!logit_map || !inds || !stats || !gw2d || !glogit_map || N < 1 || H < 1 || W < 1 || (mode != 0 && mode != 1) ||
      ((gx3d != nullptr) && (!noc_map || !dim || !gnoc_map)))
    return fail(EPROPNP_EINVAL, "prepare_dense_backward: bad argument");
  const PrepLayout L = {inds, H * W, W};
  const size_t plane = (size_t)H * W * sizeof(float);
  if (launch_fill_u32(glogit_map, 0u, (size_t)B * 2 * plane / 4, st) != EPROPNP_OK) return fail(EPROPNP_ELAUNCH, "prepare_dense_backward: memset");
  if (gx3d != nullptr && launch_fill_u32(gnoc_map, 0u, (size_t)B * 3 * plane / 4, st) != EPROPNP_OK)
    return fail(EPROPNP_ELAUNCH, "prepare_dense_backward: memset");
  PNP_LAUNCH(prepare_backward_kernel, dim3(padded_object_grid(B)), dim3(prepare_threads(N)), 0, st, noc_map, dim, logit_map,
             scale, L, stats, gx3d, gw2d, B, N, mode, gnoc_map, gdim, glogit_map, gscale);
  return check_launch("prepare_backward_kernel (dense)");
}

int launch_rslm_draw(const float* w2d, int B, int N, int P, int n_pts, unsigned long long seed, unsigned long long offset,
                     long long* inds, hipStream_t st) {
  if (B <= 0 || P <= 0 || n_pts <= 0) return EPROPNP_OK;
  if (!w2d || !inds || N < 1) return fail(EPROPNP_EINVAL, "rslm_draw: bad argument");
  if ((size_t)N * 4 > 64 * 1024) return fail(EPROPNP_EINVAL, "rslm_draw: num_pts %d too large", N);
  PNP_LAUNCH(rslm_draw_kernel, dim3((unsigned)(P * B)), dim3(64), sizeof(float) * (size_t)N, st, w2d, B, N, P, n_pts, seed,
             offset, inds);
  return check_launch("rslm_draw_kernel");
}

int launch_adaptive_delta(const float* x2d, const float* w2d, int B, int N, float rel, float* delta, float* stats,
                          hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!x2d || !w2d || !delta || !stats || N < 1) return fail(EPROPNP_EINVAL, "adaptive_delta: bad argument");
  int threads = 64;
  while (threads < 256 && threads * 4 < N) threads *= 2;
  PNP_LAUNCH(adaptive_delta_kernel, dim3(padded_object_grid(B)), dim3(threads), 0, st, x2d, w2d, B, N, rel, delta, stats);
  return check_launch("adaptive_delta_kernel");
}

// Fill `words` 32-bit words at p with v, as a KERNEL.  Every stream-ordered reset inside this library goes through it instead
// of hipMemsetAsync: captured into a hipGraph, a memset node of this ROCm stack (7.0 user space) writes garbage once eager
// launches have run between two replays (tools/ubench/graph_memset_node.py) -- a kernel node replays correctly.
__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned* __restrict__ p, unsigned v, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int launch_fill_u32(void* p, unsigned v, size_t words, hipStream_t st) {
  if (words == 0) return EPROPNP_OK;
  if (p == nullptr || ((size_t)p & 3u) != 0) return fail(EPROPNP_EINVAL, "fill: NULL or unaligned buffer");
  size_t blocks = (words + 1023) / 1024;        // four words per thread
  if (blocks > 2048) blocks = 2048;
  PNP_LAUNCH(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)p, v, words);
  return check_launch("fill_u32_kernel");
}

int launch_mc_loss_forward(const float* logw, const float* ct, int S, int B, float* loss, float* lse, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!logw || !loss || !lse || S < 1) return fail(EPROPNP_EINVAL, "mc_loss_forward: bad argument");
  PNP_LAUNCH(mc_loss_forward_kernel, dim3((B + kLossCols - 1) / kLossCols), dim3(512), 0, st, logw, ct, S, B, loss, lse);
  return check_launch("mc_loss_forward_kernel");
}

int launch_mc_loss_backward(const float* logw, const float* lse, const float* loss, const float* g, int S, int B,
                            float* glogw, float* gct, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!logw || !lse || !loss || !g || !glogw) return fail(EPROPNP_EINVAL, "mc_loss_backward: NULL pointer");
  (void)loss;
  {
    const size_t total = (size_t)S * B;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    PNP_LAUNCH(mc_loss_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, st, logw, lse, g, (const float*)nullptr,
               (const float*)nullptr, (const float*)nullptr, S, B, glogw, gct);
  }
  return check_launch("mc_loss_backward_kernel");
}

// ObjectExchange's send buffer in one launch (include/epropnp_hip.h: epropnp_exchange_pack).  Block 0 also produces the scalars:
// with sum_src the first one is a fixed-order single-workgroup sum (the Det head's norm_factor input).
__global__ __launch_bounds__(256) void exchange_pack_kernel(const float* __restrict__ rows, size_t row_floats,
                                                            const float* __restrict__ scalars, int n_scal,
                                                            const float* __restrict__ sum_src, size_t sum_floats, float sum_scale,
                                                            const float* __restrict__ row_w, int row_len, float* __restrict__ send) {
  __shared__ float red[4];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_floats; i += (size_t)gridDim.x * blockDim.x)
    send[(size_t)n_scal + i] = rows[i];
  if (blockIdx.x != 0) return;
  if (sum_src != nullptr) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};       // four independent chains per thread, combined in a fixed order
    const size_t T = blockDim.x;
    auto at = [&](size_t j) { return row_w != nullptr ? sum_src[j] * row_w[j / (size_t)row_len] : sum_src[j]; };
    size_t i = threadIdx.x;
    for (; i + 3 * T < sum_floats; i += 4 * T) {
      acc[0] += at(i); acc[1] += at(i + T); acc[2] += at(i + 2 * T); acc[3] += at(i + 3 * T);
    }
    for (; i < sum_floats; i += T) acc[0] += at(i);
    float v[1] = {(acc[0] + acc[1]) + (acc[2] + acc[3])};
    block_sum<1>(v, red);
    if (threadIdx.x == 0) send[0] = v[0] * sum_scale;
  }
  for (int i = (int)threadIdx.x + (sum_src != nullptr ? 1 : 0); i < n_scal; i += (int)blockDim.x) send[i] = scalars[i];
}

int launch_exchange_pack(const float* rows, size_t row_floats, const float* scalars, int n_scal, const float* sum_src,
                         size_t sum_floats, float sum_scale, const float* row_w, int row_len, float* send, hipStream_t st) {
  if (row_w != nullptr && (row_len < 1 || sum_src == nullptr)) return fail(EPROPNP_EINVAL, "exchange_pack: row weights need sum_src and row_len >= 1");
  if (!send || (row_floats > 0 && !rows) || n_scal < 0) return fail(EPROPNP_EINVAL, "exchange_pack: NULL pointer");
  if (sum_src != nullptr && n_scal < 1) return fail(EPROPNP_EINVAL, "exchange_pack: a sum needs a scalar slot");
  if (scalars == nullptr && n_scal > (sum_src != nullptr ? 1 : 0)) return fail(EPROPNP_EINVAL, "exchange_pack: scalars NULL");
  size_t blocks = (row_floats + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  PNP_LAUNCH(exchange_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, rows, row_floats, scalars, n_scal, sum_src, sum_floats,
             sum_scale, row_w, row_len, send);
  return check_launch("exchange_pack_kernel");
}

int launch_mc_loss_reduce(const float* loss, const float* weight, int B, float scale, float momentum, const float* nf_in,
                          int nf_count, long long nf_stride, float* nf, float* out, hipStream_t st) {
  if (B < 0) return fail(EPROPNP_EINVAL, "mc_loss_reduce: negative num_obj");
  if (!out || (B > 0 && !loss)) return fail(EPROPNP_EINVAL, "mc_loss_reduce: NULL pointer");
  if (nf_in != nullptr && nf == nullptr) return fail(EPROPNP_EINVAL, "mc_loss_reduce: norm_factor_in without norm_factor");
  if (nf_in != nullptr && (nf_count < 1 || (nf_count > 1 && nf_stride < 1))) return fail(EPROPNP_EINVAL, "mc_loss_reduce: bad norm_factor_in count / stride");
  const float one_minus_m = (float)(1.0 - (double)momentum);       // the reference's `1 - self.momentum`, rounded once
  PNP_LAUNCH(mc_loss_reduce_kernel, dim3(1), dim3(B > 4096 ? 1024 : 256), 0, st, loss, weight, B, scale, one_minus_m, momentum,
             nf_in, nf_count, nf_stride, nf, out);
  return check_launch("mc_loss_reduce_kernel");
}

int launch_mc_loss_reduce_backward(const float* logw, const float* lse, const float* weight, const float* coef,
                                   const float* gout, int S, int B, float* glogw, float* gct, hipStream_t st) {
  if (B <= 0) return EPROPNP_OK;
  if (!logw || !lse || !coef || !gout || !glogw) return fail(EPROPNP_EINVAL, "mc_loss_reduce_backward: NULL pointer");
  const size_t total = (size_t)S * B;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  PNP_LAUNCH(mc_loss_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, st, logw, lse, (const float*)nullptr, gout, coef,
             weight, S, B, glogw, gct);
  return check_launch("mc_loss_backward_kernel (reduced loss)");
}

// ----------------------------------------------------------------------------------------------------------
int launch_normal_equations(const epropnp_problem* prob, const float* pose, int clip_jac, float* jtj, float* jtr,
                            float* cost, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose || !jtj || !jtr || !cost) return fail(EPROPNP_EINVAL, "normal_equations: NULL pointer");
  const Problem d = to_device_problem(prob);
  const dim3 grid(padded_object_grid(d.B));
  if (d.N > kMaxResidentPoints || d.N == 0) {     // streaming loop, 8 waves per object (no points: it writes zeros)
    dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
      PNP_LAUNCH((normal_equations_kernel<decltype(DOF)::value, 0, decltype(BND)::value, 8>), grid, dim3(512),
                 sizeof(float) * NormalEq<decltype(DOF)::value>::NV * 16, st, d, pose, clip_jac, jtj, jtr, cost);
      return 0;
    });
    return check_launch("normal_equations_kernel (streaming)");
  }
  // one sweep, latency matters more than anything: the fewest waves that hold the points, as in lm_solve_kernel.
  // (Two objects per wave, the second one's loads in flight while the first computes, was built and measured at C2:
  //  22.9 us against 21.3 us -- 190 VGPRs halve the occupancy and cost more than the overlap gains.)
  Shape s = choose_shape(d.B, d.N, /*max_ppl=*/8, /*want_waves_total=*/0);
  int ov[2];
  if (tune_ints("ne_shape", ov, 2) && valid_shape_override(ov[0], ov[1], d.N)) { s.waves = ov[0]; s.ppl = ov[1]; }
  dispatch_shape(prob->dof, s.ppl, has_bounds(prob), s.waves, [&](auto DOF, auto PPL, auto BND, auto MAXW) -> int {
    PNP_LAUNCH((normal_equations_kernel<decltype(DOF)::value, decltype(PPL)::value, decltype(BND)::value, decltype(MAXW)::value>),
               grid, dim3(64 * s.waves),
               sizeof(float) * (decltype(MAXW)::value <= 4 ? s.waves * kSumTStride<NormalEq<decltype(DOF)::value>::NV>
                                                           : NormalEq<decltype(DOF)::value>::NV * 16),
               st, d, pose, clip_jac, jtj, jtr, cost);
    return 0;
  });
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

int launch_cost_pose_cam_grad(const epropnp_problem* prob, const float* poses, const float* weights, int num_poses,
                              int m_pose, float* out_m, float* out_gk, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!poses || (!out_m && !out_gk) || num_poses < 1 || m_pose >= num_poses)
    return fail(EPROPNP_EINVAL, "cost_pose_cam_grad: NULL pointer / bad pose index");
  const Problem d = to_device_problem(prob);
  const dim3 grid(padded_object_grid(d.B)), block(d.N > 128 ? 256 : (d.N > 64 ? 128 : 64));
  dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
    PNP_LAUNCH((cost_pose_cam_grad_kernel<decltype(DOF)::value, decltype(BND)::value>), grid, block, 0, st, d, poses, weights,
               num_poses, m_pose, out_m, out_gk);
    return 0;
  });
  return check_launch("cost_pose_cam_grad_kernel");
}

}  // namespace pnp
