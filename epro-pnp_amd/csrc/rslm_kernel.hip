// rslm_kernel.hip -- the random-sample LM initialiser (RSLMSolver.solve) as ONE kernel.
//
// Replaces epropnp/levenberg_marquardt.py:283-353: center_based_init, the weighted sub-sampling without replacement
// (torch.multinomial), the random rotations, P x B small LM solves on the sub-samples, the full-set cost of every
// proposal and the argmin -- ~100 ATen launches and (P,B,n,.) gathered copies in the reference (and ~80 in this
// package's composite path), none of which does enough work to fill the GPU at the detection shape (600 objects x
// 128 points x 64 proposals).
//
// One 256-thread workgroup owns one object: its N correspondences are staged once in LDS; the 16 DPP rows of the
// workgroup each run one proposal at a time (16 lanes = the <= 16 sub-sampled points): draw the indices (exponential
// race keys in LDS, row-wide argmin per pick: the same law as torch.multinomial(replacement=False) and the same
// Philox stream as rslm_draw_kernel), solve (lm_iterate, reductions are row_ror adds), score on all N points, keep
// the best.  P / 16 rounds, then a 16-way argmin.
#include "dispatch.h"
#include "lm_core.h"
#include "pnp_host.h"
#include "tuning.h"

namespace pnp {

constexpr int kRslmMaxPts = 512;    // LDS: 32 B (point + reciprocal weight) + 16 x 4 B (keys of the 16 proposals in flight) per point
constexpr int kRslmRows = 16;       // DPP rows per workgroup

PNP_FN float row_min16(float x) { return -row_max16(-x); }

// (tuning builds: cycles of thread 0 in [staging + centre init | key draw | 16 picks | sub-sample solve | scoring | argmin] -- PNP_PHASE, tuning.h)
int tuning_rslm_phase_cycles(unsigned long long* out, int reset) { return tuning::read_cycles(out, reset, false); }

template <int DOF, bool BOUNDS>
__global__ __launch_bounds__(256) void rslm_solve_kernel(
Problem p, LmParams lm, int P, int n_pts, unsigned long long seed,
                                                          unsigned long long offset_in,
                                                          const unsigned long long* __restrict__ offset_dev,
                                                          const long long* __restrict__ inds,
                                                          const float* __restrict__ rot, float* __restrict__ pose_out,
                                                          float* __restrict__ cost_out, int parts, float* __restrict__ cand_out, int to_cand) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NV = NormalEq<DOF>::NV;
  // parts > 1: the proposals of an object are dealt to `parts` workgroups (v = b * parts + part), each keeps the best of its
  // share and writes it to cand_out[part][b][PL + 1]; rslm_reduce_kernel picks the winner.  An object's 64 proposals are
  // four rounds of one workgroup: at 600 objects on 256 CUs a CU gets two or three such workgroups (as long as 768 objects
  // take), and at <= 256 objects the rounds in sequence are the whole run time (profiles/r03_rslm_parts.txt).
  const int v = object_of_block(p.B * parts);
  if (v >= p.B * parts) return;
  const int b = v / parts, part = v - b * parts;
  const int P_lo = (int)(((long long)P * part) / parts), P_hi = (int)(((long long)P * (part + 1)) / parts);
  const int tid = (int)threadIdx.x, l16 = tid & 15, row = tid >> 4, N = p.N;
  const unsigned long long offset = offset_in + (offset_dev ? *offset_dev : 0ull);
  PNP_PHASES_BEGIN(6);
  const int Np = (N + 3) & ~3;
  PNP_DYN_SMEM(float, smem);
  float* sX = smem;                 // [N][3]
  float* sU = sX + 3 * Np;          // [N][2]
  float* sW = sU + 2 * Np;          // [N][2]
  float* key = sW + 2 * Np;         // [16][Np]
  float* red = key + kRslmRows * Np;   // [128]
  float* sIw = red + 128;           // [N] reciprocal mean weight of a point (race_inv_weight): one multiply per key, 64 proposals

  float K[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, K, bd, delta);
  for (int n = tid; n < N; n += 256) {
    const Point q = load_point(p, b, n);
    sX[3 * n] = q.X; sX[3 * n + 1] = q.Y; sX[3 * n + 2] = q.Z;
    sU[2 * n] = q.u; sU[2 * n + 1] = q.v;
    sW[2 * n] = q.wu; sW[2 * n + 1] = q.wv;
    sIw[n] = race_inv_weight(0.5f * (q.wu + q.wv));
  }
  __syncthreads();

  // ---- center_based_init (:283-298): t0 = [mean(rays), 1] * depth, rays = dehomogenised K^-1 [x2d, 1] ----
  float t0[3];
  {
    const float c00 = K[4] * K[8] - K[5] * K[7], c01 = K[2] * K[7] - K[1] * K[8], c02 = K[1] * K[5] - K[2] * K[4];
    const float c10 = K[5] * K[6] - K[3] * K[8], c11 = K[0] * K[8] - K[2] * K[6], c12 = K[2] * K[3] - K[0] * K[5];
    const float c20 = K[3] * K[7] - K[4] * K[6], c21 = K[1] * K[6] - K[0] * K[7], c22 = K[0] * K[4] - K[1] * K[3];
    const float idet = 1.0f / (K[0] * c00 + K[1] * c10 + K[2] * c20);
    auto ray = [&](int n, float& rx, float& ry) {
      const float u = sU[2 * n], v = sU[2 * n + 1];
      const float hx = (c00 * u + c01 * v + c02) * idet, hy = (c10 * u + c11 * v + c12) * idet;
      const float hz = fmaxf((c20 * u + c21 * v + c22) * idet, 1e-6f);
      rx = hx / hz;
      ry = hy / hz;
    };
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = tid; n < N; n += 256) {
      float rx, ry;
      ray(n, rx, ry);
      s[0] += rx; s[1] += ry; s[2] += sX[3 * n]; s[3] += sX[3 * n + 1]; s[4] += sX[3 * n + 2];
    }
    block_sum<5>(s, red);
    float mean[5], d2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 5; ++i) mean[i] = s[i] / (float)N;
    for (int n = tid; n < N; n += 256) {
      float rx, ry;
      ray(n, rx, ry);
      const float e[5] = {rx - mean[0], ry - mean[1], sX[3 * n] - mean[2], sX[3 * n + 1] - mean[3], sX[3 * n + 2] - mean[4]};
#pragma unroll
      for (int i = 0; i < 5; ++i) d2[i] = fmaf(e[i], e[i], d2[i]);
    }
    block_sum<5>(d2, red);
    float var[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) var[i] = d2[i] / (float)(N - 1);      // unbiased, as torch.std
    float depth;
    if (DOF == 4) {
      depth = sqrtf(var[3]) / fmaxf(sqrtf(var[1]), 1e-6f);
    } else {
      depth = 0.816496580927726f * sqrtf(var[2] + var[3] + var[4]) / fmaxf(sqrtf(var[0] + var[1]), 1e-6f);
    }
    t0[0] = mean[0] * depth; t0[1] = mean[1] * depth; t0[2] = depth;
  }

  PNP_PHASE(0);
  float Kv[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Kv[i] = to_vgpr(K[i]);
  const float delta_v = to_vgpr(delta), zmin_v = to_vgpr(p.z_min), inv_eps_v = to_vgpr(p.inv_huber_eps);

  float best_cost = INFINITY, best_pose[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) best_pose[i] = 0.f;
  float* mykey = key + row * Np;
  for (int j0 = P_lo; j0 < P_hi; j0 += kRslmRows) {
    const bool active = (j0 + row) < P_hi;
    const int j = active ? j0 + row : P_hi - 1;
    const size_t prow = (size_t)j * p.B + b;                 // (proposal, object) row of the composite path

    // ---- sub-sample: n_pts distinct indices ~ mean weight (:305-308) ----
    int my_idx = -1;
    if (inds != nullptr) {
      if (l16 < n_pts) my_idx = (int)inds[prow * n_pts + l16];
    } else {
      // exponential-race keys, packed with the point index (pnp_math.h: race_key); same Philox stream and the same
      // winners as rslm_draw_kernel.  N <= 128: each lane sorts its 8 keys in registers once (19 compare-exchanges) and
      // a pick is a row-wide min over the heads + a conditional shift; larger N scans the keys in LDS per pick.
      unsigned* ukey = reinterpret_cast<unsigned*>(mykey);
      for (int n4 = l16; 4 * n4 < N; n4 += 16) {
        const Philox4 r = philox4x32_10((uint32_t)prow, (uint32_t)n4, (uint32_t)offset, (uint32_t)(offset >> 32),
                                        (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5bd1e995u);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 4 * n4 + q;
          if (n < N) ukey[n] = race_key(r.v[q], sIw[n], n);
        }
      }
      wave_lds_fence();
      PNP_PHASE(1);
      if (N <= 128) {
        unsigned k8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int n = l16 + 16 * i;
          k8[i] = (n < N) ? ukey[n] : 0xffffffffu;
        }
        auto cx = [&](int i, int j) {
          const unsigned lo = min(k8[i], k8[j]), hi = max(k8[i], k8[j]);
          k8[i] = lo; k8[j] = hi;
        };
        cx(0, 1); cx(2, 3); cx(4, 5); cx(6, 7); cx(0, 2); cx(1, 3); cx(4, 6); cx(5, 7); cx(1, 2); cx(5, 6); cx(0, 4);
        cx(3, 7); cx(1, 5); cx(2, 6); cx(1, 4); cx(3, 6); cx(2, 4); cx(3, 5); cx(3, 4);
        for (int k = 0; k < n_pts; ++k) {
          const unsigned m = row_min16_u32(k8[0]);
          const bool own = (k8[0] == m) && (m != 0xffffffffu);
          const int wi = (m < kRaceInf) ? (int)(m & kRaceIdxMask) : (k % N);   // fewer positive weights than n_pts
          if (l16 == k) my_idx = wi;
#pragma unroll
          for (int i = 0; i < 7; ++i) k8[i] = own ? k8[i + 1] : k8[i];
          k8[7] = own ? 0xffffffffu : k8[7];
        }
      } else {
        for (int k = 0; k < n_pts; ++k) {
          unsigned best = 0xffffffffu;
          for (int n = l16; n < N; n += 16) best = min(best, ukey[n]);
          const unsigned m = row_min16_u32(best);
          const int slot = (int)(m & kRaceIdxMask);
          const int wi = (m < kRaceInf) ? slot : (k % N);
          if (l16 == k) my_idx = wi;
          if (m != 0xffffffffu && l16 == (slot & 15)) ukey[slot] = 0xffffffffu;
          wave_lds_fence();
        }
      }
    }
    PNP_PHASE(2);
    Point pt;
    if (my_idx >= 0 && my_idx < N) {
      pt.X = sX[3 * my_idx]; pt.Y = sX[3 * my_idx + 1]; pt.Z = sX[3 * my_idx + 2];
      pt.u = sU[2 * my_idx]; pt.v = sU[2 * my_idx + 1];
      pt.wu = sW[2 * my_idx]; pt.wv = sW[2 * my_idx + 1];
    } else {
      pt.X = pt.Y = pt.Z = pt.u = pt.v = pt.wu = pt.wv = 0.f;
    }

    // ---- starting pose: centre-based translation + random rotation (:310-321) ----
    float pose[PL];
    pose[0] = t0[0]; pose[1] = t0[1]; pose[2] = t0[2];
    if (rot != nullptr) {
#pragma unroll
      for (int i = 3; i < PL; ++i) pose[i] = rot[prow * (PL - 3) + (i - 3)];
    } else {
      const Philox4 r = philox4x32_10((uint32_t)prow, 0xffffffffu, (uint32_t)offset, (uint32_t)(offset >> 32),
                                      (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x9e3779b9u);
      if (DOF == 4) {
        pose[3] = u01(r.v[0]) * 6.283185307179586f;
      } else {
        float g[4];
        box_muller(r.v[0], r.v[1], g[0], g[1]);
        box_muller(r.v[2], r.v[3], g[2], g[3]);
        const float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3]);
        const bool degenerate = nrm < lm.eps;
#pragma unroll
        for (int i = 0; i < 4; ++i) pose[3 + (i < PL - 3 ? i : 0)] = degenerate ? (i == 0 ? 1.f : 0.f) : g[i] / nrm;
      }
    }

    // ---- LM / GN on the sub-sample, one problem per DPP row ----
    auto sweep = [&](const float* ps, bool clip, float (&acc)[NV]) {
      float R[9], t[3];
      pose_to_rot<DOF>(ps, R);
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = ps[i];
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] = 0.f;
      point_normal_eq<DOF, BOUNDS>(pt, Kv, R, t, zmin_v, delta_v, inv_eps_v, bd, clip, acc);
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] = row_sum16(acc[i]);
    };
    float cur[NV];
    int bits = 0;
    int st_bits = 0;      // sub-sample solves may be degenerate by design (16 random points): not reported
    lm_iterate<DOF>(lm, sweep, pose, cur, bits, st_bits);
    PNP_PHASE(3);

    // ---- score the proposal on the full correspondence set (:343); hardware rcp / sqrt as in the AMIS sweeps: the
    // score only ranks proposals (and is compared with another pose's cost by LMSolver.solve), 1-ulp effects are moot
    float c = 0.f;
    {
      float R[9], KR[9], Kt[3];
      pose_to_rot<DOF>(pose, R);
      compose_kr_kt(K, R, pose, KR, Kt);
      for (int n = l16; n < N; n += 16) {
        SweepPoint sp;
        sp.X = sX[3 * n]; sp.Y = sX[3 * n + 1]; sp.Z = sX[3 * n + 2];
        sp.wu = sW[2 * n]; sp.wv = sW[2 * n + 1];
        sp.cu = -sU[2 * n] * sp.wu; sp.cv = -sU[2 * n + 1] * sp.wv;
        c += sweep_cost<BOUNDS>(sp, KR, Kt, zmin_v, delta_v, bd);
      }
      c = row_sum16(c);
    }
    PNP_PHASE(4);
    if (active && (j0 == P_lo || c < best_cost)) {
      best_cost = c;
#pragma unroll
      for (int i = 0; i < PL; ++i) best_pose[i] = pose[i];
    }
  }

  // ---- argmin over the 16 rows (ties: lowest row = lowest proposal index of the first round) ----
  __syncthreads();
  float* cand = red;      // [16][PL + 1]
  if (l16 == 0) {
    cand[row * (PL + 1)] = (row < P_hi - P_lo) ? best_cost : INFINITY;
#pragma unroll
    for (int i = 0; i < PL; ++i) cand[row * (PL + 1) + 1 + i] = best_pose[i];
  }
  __syncthreads();
  if (tid == 0) {
    int w = 0;
    float wc = cand[0];
    for (int r = 1; r < kRslmRows && r < P_hi - P_lo; ++r) {
      const float cr = cand[r * (PL + 1)];
      if (cr < wc) { wc = cr; w = r; }
    }
    if (parts > 1 || to_cand) {     // (to_cand: one part, but the consumer -- the LM launch's start selection -- reads the candidate layout)
      float* dst = cand_out + ((size_t)part * p.B + b) * (PL + 1);
      dst[0] = wc;
#pragma unroll
      for (int i = 0; i < PL; ++i) dst[1 + i] = cand[w * (PL + 1) + 1 + i];
    } else {
#pragma unroll
      for (int i = 0; i < PL; ++i) pose_out[(size_t)b * PL + i] = cand[w * (PL + 1) + 1 + i];
      if (cost_out) cost_out[b] = wc;
    }
  }
  PNP_PHASES_FLUSH(6);
}

// winner over the parts of an object (ties: the lowest part = the lowest proposal index, as the single-workgroup kernel)
// rival (optional): a given start pose and its cost -- force_init_solve=True with a pose_init takes, per object, the cheaper
// of {pose_init, RSLM pose} (levenberg_marquardt.py:124-130: `use_init = cost_init < cost_init_solve`); folded in here it
// saves the separate selection launch
template <int PL>
__global__ __launch_bounds__(256) void rslm_reduce_kernel(const float* __restrict__ cand, int B, int parts,
                                                          float* __restrict__ pose_out, float* __restrict__ cost_out,
                                                          const float* __restrict__ rival_pose,
                                                          const float* __restrict__ rival_cost) {
  const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (b >= B) return;
  int w = 0;
  float wc = cand[(size_t)b * (PL + 1)];
  for (int q = 1; q < parts; ++q) {
    const float c = cand[((size_t)q * B + b) * (PL + 1)];
    if (c < wc) { wc = c; w = q; }
  }
  const float* src = cand + ((size_t)w * B + b) * (PL + 1) + 1;
  if (rival_pose != nullptr && rival_cost[b] < wc) src = rival_pose + (size_t)b * PL;     // (cost_out keeps the RSLM cost)
#pragma unroll
  for (int i = 0; i < PL; ++i) pose_out[(size_t)b * PL + i] = src[i];
  if (cost_out) cost_out[b] = wc;
}

// Parts per object (whole rounds of 16 proposals each), measured (profiles/r03_rslm_parts.txt): up to 256 objects the kernel
// is pure latency -- rounds in sequence -- and quarters win (75 x 64 proposals: 40 -> 22 us); up to 768 objects halves even
// out the two-or-three-workgroups-per-CU quantisation (600 x 64: 71 -> 63 us); beyond that the per-workgroup staging of the
// points costs more than the balance gains.  EPROPNP_RSLM_PARTS=<n> overrides.
static int rslm_parts(int B, int P) {
  const int cus = device_cu_count();         // (256 on an MI355X; the thresholds scale with a partitioned device)
  int best = (B <= cus) ? 4 : (B <= 3 * cus ? 2 : 1);
  if (cus < 16) best = 1;                    // nothing to balance over (the tests' one-CU CPU emulation)
  while (best > 1 && P % (kRslmRows * best) != 0) best >>= 1;
  { int ov[1]; if (tune_ints("rslm_parts", ov, 1) && (ov[0] == 1 || ov[0] == 2 || ov[0] == 4) && P % (kRslmRows * ov[0]) == 0) best = ov[0]; }
  return best;
}

unsigned long long rslm_scratch_bytes(const epropnp_problem* prob, int P) {
  if (prob == nullptr || prob->num_obj <= 0 || P < 1) return 0;
  const int q = rslm_parts(prob->num_obj, P);
  const int PL = prob->dof == 6 ? 7 : 4;
  // (one part: the candidate record through which the one-call forward hands the start to the LM launch, which then also takes the
  // cheaper-of-two selection against pose_init -- no select launch)
  return sizeof(float) * (unsigned long long)q * prob->num_obj * (PL + 1);
}

int launch_rslm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, int P, int n_pts, unsigned long long seed,
                      unsigned long long offset, const unsigned long long* offset_dev, const long long* inds, const float* rot,
                      float* pose_out, float* cost_out, void* scratch, unsigned long long scratch_bytes, hipStream_t st,
                      const float* rival_pose, const float* rival_cost, bool* rival_taken, int* deferred_parts) {
  // deferred_parts != nullptr (and a scratch that holds the candidates): the reduce launch is LEFT OUT and *deferred_parts = parts
  // (>= 1) tells the caller to hand `scratch` to the LM launch as its start selection (lm_core.h: StartSelect); pose_out / cost_out
  // are then not written.  *deferred_parts = 0: pose_out holds the start as usual.
  if (deferred_parts) *deferred_parts = 0;
  if (rival_taken) *rival_taken = false;
  if (int rc = check_problem(prob)) return rc;
  if (!lm) return fail(EPROPNP_EINVAL, "rslm_solve: params NULL");
  if (prob->num_obj == 0) return EPROPNP_OK;
  if (!pose_out) return fail(EPROPNP_EINVAL, "rslm_solve: NULL pose pointer");
  if (P < 1 || n_pts < 1 || n_pts > 16)
    return fail(EPROPNP_EINVAL, "rslm_solve: need num_proposals >= 1 and 1 <= num_points <= 16 (got %d, %d)", P, n_pts);
  if (prob->num_pts < 2 || prob->num_pts > kRslmMaxPts)
    return fail(EPROPNP_EINVAL, "rslm_solve: num_pts %d outside [2, %d]", prob->num_pts, kRslmMaxPts);
  if (lm->num_iter < 0 || lm->num_iter > 31 * 1000) return fail(EPROPNP_EINVAL, "rslm_solve: bad num_iter");
  const Problem d = to_device_problem(prob);
  LmParams k;
  k.num_iter = lm->num_iter; k.fast_mode = lm->fast_mode;
  k.min_diag = lm->min_lm_diagonal; k.max_diag = lm->max_lm_diagonal;
  k.min_rel_decrease = lm->min_relative_decrease; k.radius0 = lm->initial_trust_region_radius;
  k.radius_max = lm->max_trust_region_radius; k.eps = lm->eps; k.split_timeout = 0;
  const int Np = (d.N + 3) & ~3;
  const size_t smem = sizeof(float) * ((size_t)(8 + kRslmRows) * Np + 128);
  int parts = rslm_parts(d.B, P);
  const int PLh = prob->dof == 6 ? 7 : 4;
  if (parts > 1 && (scratch == nullptr || scratch_bytes < sizeof(float) * (size_t)parts * d.B * (PLh + 1))) parts = 1;
  const bool to_cand = deferred_parts != nullptr && scratch != nullptr && scratch_bytes >= sizeof(float) * (size_t)parts * d.B * (PLh + 1);
  const dim3 grid(padded_object_grid(d.B * parts)), block(256);
  dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
    PNP_LAUNCH((rslm_solve_kernel<decltype(DOF)::value, decltype(BND)::value>), grid, block, smem, st, d, k, P, n_pts, seed,
               offset, offset_dev, inds, rot, pose_out, cost_out, parts, (float*)scratch, (int)to_cand);
    return 0;
  });
  if (to_cand) {
    *deferred_parts = parts;
    if (rival_taken) *rival_taken = rival_pose != nullptr && rival_cost != nullptr;
    return check_launch("rslm_solve_kernel");
  }
  if (parts > 1) {
    if (int rc = check_launch("rslm_solve_kernel")) return rc;
    const dim3 rgrid((d.B + 255) / 256);
    if (prob->dof == 6) {
      PNP_LAUNCH((rslm_reduce_kernel<7>), rgrid, block, 0, st, (const float*)scratch, d.B, parts, pose_out, cost_out, rival_pose,
                 rival_cost);
    } else {
      PNP_LAUNCH((rslm_reduce_kernel<4>), rgrid, block, 0, st, (const float*)scratch, d.B, parts, pose_out, cost_out, rival_pose,
                 rival_cost);
    }
    if (rival_taken) *rival_taken = rival_pose != nullptr && rival_cost != nullptr;
  }
  return check_launch("rslm_solve_kernel");
}

}  // namespace pnp
