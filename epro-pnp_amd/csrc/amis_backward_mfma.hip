// amis_backward_mfma.hip -- backward of the AMIS log-weights with the pose x point projection on the matrix cores.
//
// Same contract as amis_backward_kernel (amis_kernels.hip): gradient of
//     sum_j a_j * cost(pose_j; x3d, x2d, w2d, delta),   a_j = -g_logw[j]  (+ g_init * cost(pose_init))
// w.r.t. the correspondences, recomputed from the points (the reference replays autograd through evaluate_pnp,
// epropnp/common.py:67-100, camera.py:21-30,81-93, cost_fun.py:45-61).  What differs is where the arithmetic runs:
//   * forward projection  h = (K R | K t)(X,Y,Z,1)^T of 16 poses x 16 points: three MFMAs (A = pose rows from LDS, B = point
//     tile in registers) -- v_mfma_f32_16x16x4_f32 as in amis_forward_mfma.hip, or (BF16, the default for <= 4 resident
//     point tiles) ONE v_mfma_f32_16x16x32_bf16 per image row on bf16x3-split operands: fp32-level accuracy at 55 % of the
//     matrix time (wave_ops.h: ProjOp; the points are split once per chunk, the pose rows on the fly, 21 instructions per
//     pose tile);
//   * back-projection  g_x3d[n] += sum_j (K R)_j^T g_h[j,n] stays on the VALU (9 FMAs per pair).  It IS expressible as
//     the same MFMA with the roles swapped (A = the per-pair g_h values, which the forward MFMA leaves in exactly the
//     lane layout an A operand needs; B = pose rows indexed by output coordinate), and that variant was built and
//     measured: 12 more MFMAs per 16x16 tile for 3 useful output columns of 16, each waiting on VALU results and on
//     the previous accumulate -- 1.6-1.9 ms instead of 1.08 ms at C2 (profiles/r01_bwd_mfma_sweep.txt).  Dropped.  Round 6 built it
//     again on v_mfma_f32_4x4x1_16b_f32 (2 passes; a block of four lanes = four points, no wasted columns to speak of): 31 % fewer
//     packed instructions, 32 VGPRs less, 924-948 us against 780-793 -- vector arithmetic does not run underneath an MFMA on this
//     part, of its own wave or of a neighbour (profiles/r06_mfma_valu_overlap.txt, tools/ubench/mfma_4x4x1_layout.hip).  Dropped.
//   * the VALU keeps what is genuinely per pair: perspective divide, weighted residual, Huber weight, and the
//     accumulation of the gradients -- since round 6 on explicit 2-vectors: two point-poses per v_pk_*_f32 (22.6 wave-level
//     instructions per pair with four resident tiles, 2 of them transcendental; the loop is bound by how often a wave gets to issue, profiles/r06_bwd_packed.txt).
// The weighted poses are built ONCE into an LDS table (compacted: the low-weight tail whose total |weight| is below
// drop_eps -- default 2^-24 -- of the object's total is dropped before tiling, mass_drop_threshold in amis_common.h;
// EPROPNP_BWD_DROP=0 keeps every non-zero sample), then every wave sweeps all pose tiles for its own points.
#include "amis_common.h"
#include "dispatch.h"
#include "tuning.h"

namespace pnp {

// Register budget (the packed pair loop of round 6 with pair accumulators, compiled WITHOUT the SLP vectoriser -- build.py): <= 2
// resident point tiles fit three waves per SIMD (bf16 projection: 152 VGPRs), four resident tiles are compiled for two (248 / 256
// VGPRs with the projection clamp, software pipeline over the tiles included) -- and are the faster shape at C2 all the same (launcher comment).
// Why no vectoriser: the packed fp32 instructions it forms include shapes that return wrong results on the MI355X while a bf16 MFMA of a
// neighbouring wave executes (profiles/r05_pk_opsel_erratum.txt) -- the run-to-run different gradients of round 5.  The packed
// arithmetic of this kernel is written by hand, on explicit 2-vectors, in shapes that the erratum does not touch (see the pair loop).
// Rules that stay: no scratch access inside a loop of this kernel (tools/scratch_audit.py --check), no packed instruction of the
// known-bad shape in ANY function of the library (tools/pk_opsel_fix.py --audit, run by the build), and every instantiation passes the
// repeated-launch test at full occupancy (tests/test_determinism_gpu.py) -- a defect of this kind is invisible to a tolerance.
template <int DOF, bool BOUNDS, int NPT, bool BF16>
constexpr int bwd_min_waves() { return (NPT <= 2) ? PNP_BWD_MINW : PNP_BWD_MINW4; }

template <int DOF, bool BOUNDS, int NPT, bool BF16 = false>
__global__ __launch_bounds__(512, (bwd_min_waves<DOF, BOUNDS, NPT, BF16>())) void amis_backward_mfma_kernel(Problem p, const float* __restrict__ pose_samples,
                                                                     const float* __restrict__ g_logw, int S,
                                                                     const float* __restrict__ pose_init,
                                                                     const float* __restrict__ g_init, int P16,
                                                                     float* __restrict__ gx3d, float* __restrict__ gx2d,
                                                                     float* __restrict__ gw2d, float* __restrict__ gdelta,
                                                                     int nsplit, float drop_eps, int gw_rows) {
  constexpr int PL = PoseLen<DOF>::value;
  typedef ProjOp<BF16> Proj;      // the projection MFMA: fp32 16x16x4, or the bf16x3 split on 16x16x32 (wave_ops.h)
  // nsplit > 1 (few objects): an object's point chunks are dealt to nsplit workgroups (v = b * nsplit + part), each with
  // its own copy of the pose table; per-point gradients are disjoint, grad_delta comes out as nsplit partials per object
  const int v = object_of_block(p.B * nsplit);
  if (v >= p.B * nsplit) return;
  const int b = v / nsplit, part = v - b * nsplit;
  const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id(), W = T >> 6;

  PNP_PHASES_BEGIN(6);      // (tuning builds: cycles in [weights | drop threshold | compaction | pose rows | sweep + outputs | tail])
  PNP_DYN_SMEM(float, smem);
  float* ptab = smem;                                   // [P16 / 4][12][4]  (K R | K t) of the compacted poses, four poses per component (below)
  float* wtab = ptab + 12 * P16;                        // [P16]      weights of the compacted poses
  float* wraw = wtab + P16;                             // [P16]      weights by sample index (0 = dropped)
  int* idx = reinterpret_cast<int*>(wraw + P16);        // [P16]      sample index of compacted pose c
  float* red = reinterpret_cast<float*>(idx + P16);     // [80]       reductions, lane counts, active count
  float* hist = red + 80;                               // [kDropHistFloats] weight histogram of the drop threshold
  // [gw_rows][2] this object's grad_w2d, parked until the threshold's gradient (one number per object, known only after the
  // last chunk) can be added on the way out: the fold below then costs no second pass over grad_w2d in global memory
  float* gwl = hist + kDropHistFloats;
  const bool fold = (p.delta_stats != nullptr) && (nsplit == 1);
  const bool park = fold && (gw_rows >= p.N);

  float Kc[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, Kc, bd, delta);
  // Residuals in units of delta (weights pre-multiplied by 1 / delta, huber_scale): min(rho, delta) becomes the clamp
  // modifier of a full-rate multiply (sat_mul); the Huber weight `coef` is unchanged, the accumulators come out scaled
  // by 1 / delta (A1, d/d delta) or 1 / delta^2 (A2, the back-projection) and are rescaled once per point at the end.
  const HuberScale hs = huber_scale(delta);
  const float zmin_v = to_vgpr(p.z_min), one_v = to_vgpr(1.0f);
  // "in front of the depth clamp" as a 0/1 factor from ONE full-rate instruction: clamp((h_z - z_lo) * 2^60, 0, 1) with z_lo
  // the float just below z_min is 1 exactly when h_z >= z_min (compare + two selects issue at half rate, tools/ubench)
  const float front_scale = to_vgpr(0x1p60f);
  const float front_off = to_vgpr(-nextafterf(p.z_min, -1.0f) * 0x1p60f);
  const float tiny_v = to_vgpr(1e-30f);     // keeps rsq finite at a zero residual; folded into the norm's first fma

  const bool with_init = (pose_init != nullptr) && (g_init != nullptr);
  const int P = S + (with_init ? 1 : 0);      // pose index S = pose_init
  // g_init[b] is wave-uniform, i.e. a SCALAR load: inside the per-lane (m >= S) arm below the compiler does not branch
  // around it when no lane takes the arm (scalar loads ignore EXEC), and a NULL g_init faulted.  Fetch it under a
  // uniform branch instead.
  const float g_init_b = with_init ? g_init[b] : 0.f;

  // ---- weights, drop threshold (mass_drop_threshold, amis_common.h), compaction ----
  float amax = 0.f;
  for (int m = tid; m < P; m += T) {
    const float w = (m < S) ? -g_logw[(size_t)m * p.B + b] : g_init_b;        // logw = -cost - const
    wraw[m] = w;
    if (m < S) amax = fmaxf(amax, fabsf(w));
  }
  amax = block_max(amax, red);          // (barriers inside: wraw is visible to every wave afterwards)
  __syncthreads();
  PNP_PHASE(0);
  const float askip = mass_drop_threshold([&](int m) { return fabsf(wraw[m]); }, S, amax, drop_eps, hist);
  for (int m = tid; m < S; m += T)
    if (fabsf(wraw[m]) <= askip) wraw[m] = 0.f;
  __syncthreads();
  PNP_PHASE(1);
  if (wv == 0) {     // ordered compaction by one wave: lane l owns the contiguous samples [l*seg, (l+1)*seg)
    const int seg = (P + 63) >> 6;
    const int m0 = lane * seg, m1 = min(P, m0 + seg);
    int cnt = 0;
    for (int m = m0; m < m1; ++m) cnt += (wraw[m] != 0.f) ? 1 : 0;
    int* lcnt = reinterpret_cast<int*>(red);
    lcnt[lane] = cnt;
    wave_lds_fence();
    int off = 0, total = 0;
    for (int l = 0; l < 64; ++l) {
      const int c = lcnt[l];
      off += (l < lane) ? c : 0;
      total += c;
    }
    for (int m = m0; m < m1; ++m)
      if (wraw[m] != 0.f) idx[off++] = m;
    if (lane == 0) lcnt[64] = total;
  }
  __syncthreads();
  PNP_PHASE(2);
  const int nact = reinterpret_cast<const int*>(red)[64];
  const int ntile = (nact + 15) >> 4;
  // Pose table layout: [group of 4 poses][component row * 4 + k][pose in group] -- component (row, k) of (K R | K t), k = 3 the
  // translation column.  The sweep's lanes each own one group (poses g4 .. g4 + 3 of a tile): a float4 read is one component of the
  // lane's four poses, i.e. two aligned register PAIRS (poses 0, 1 | 2, 3) -- the operands of the packed pair loop below.
  for (int c = tid; c < ntile * 16; c += T) {
    float* dst = ptab + (c >> 2) * 48 + (c & 3);
    if (c < nact) {
      const int m = idx[c];
      const float* src = (m < S) ? pose_samples + ((size_t)m * p.B + b) * PL : pose_init + (size_t)b * PL;
      float ps[PL], R[9], KR[9], Kt[3];
#pragma unroll
      for (int i = 0; i < PL; ++i) ps[i] = src[i];
      pose_to_rot<DOF>(ps, R);
      compose_kr_kt(Kc, R, ps, KR, Kt);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        dst[(4 * r + 0) * 4] = KR[3 * r]; dst[(4 * r + 1) * 4] = KR[3 * r + 1]; dst[(4 * r + 2) * 4] = KR[3 * r + 2];
        dst[(4 * r + 3) * 4] = Kt[r];
      }
      wtab[c] = wraw[m];
    } else {     // padding of the last tile: a harmless pose (depth 1) with zero weight
#pragma unroll
      for (int k = 0; k < 12; ++k) dst[k * 4] = (k == 11) ? 1.f : 0.f;
      wtab[c] = 0.f;
    }
  }
  __syncthreads();
  PNP_PHASE(3);

  const int col = lane & 15, kk = lane >> 4, g4 = kk * 4;
  float gd = 0.f;
  const int chunk_pts = W * NPT * 16;
  for (int c0 = part * chunk_pts; c0 < p.N; c0 += nsplit * chunk_pts) {
    // this wave's point tiles q = wv + W * i of the chunk; lane = (point column, k)
    typename Proj::T rB[NPT];
    float4 rW[NPT];
    // per resident tile: the four sums behind d/du, d/dw and the back-projected gradient, as PAIRS over the lane's poses (0, 2 | 1, 3):
    // folded once per chunk (below).  14 VGPRs per tile instead of 7 -- the two-tile instantiation has them to spare below the
    // three-waves-per-SIMD budget of 168, and they save 3.5 wave-level instructions per pair (8 scalar fmacs -> 4 packed fmas per two
    // pairs, no folding adds per pose tile).
    f32x2 A1x[NPT], A1y[NPT], A2x[NPT], A2y[NPT];
    f32x2 gXv[NPT], gYv[NPT], gZv[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const Point q = load_point(p, b, c0 + (wv + W * i) * 16 + col);      // zero weight beyond N
      rB[i] = Proj::b((kk == 0) ? q.X : (kk == 1) ? q.Y : (kk == 2) ? q.Z : 1.0f);
      const float wu = q.wu * hs.inv_delta, wv = q.wv * hs.inv_delta;
      rW[i] = make_float4(wu, -q.u * wu, wv, -q.v * wv);      // (w_u, c_u | w_v, c_v): the factors in the low halves of their pairs
      A1x[i] = A1y[i] = A2x[i] = A2y[i] = f32x2{0.f, 0.f};
      gXv[i] = gYv[i] = gZv[i] = f32x2{0.f, 0.f};
    }
    f32x2 gsat2 = {0.f, 0.f};   // sum_pairs a_j min(|r|^2, 1): the second term of d/d delta (below), poses (0, 2) | (1, 3) of the lanes
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x2 tiny2 = {tiny_v, tiny_v};
    // The pair loop on 2-VECTORS: a lane's four poses of a tile are two register pairs (MFMA result elements 0, 1 | 2, 3; the pose
    // table above delivers every per-pose operand as such pairs), and every multiply / fma below is ONE v_pk_*_f32 for two pairs.  Packed
    // fp32 issues at half the rate of the scalar form, so the arithmetic throughput is the same -- but this loop is not bound by that:
    // it is bound by how often a wave gets to issue (profiles/r06_bwd_probe.txt: 10 % fewer instructions changed nothing, a fourth wave
    // per SIMD gave 6 %), and the packed form needs ~40 % fewer issues per pair.  Shapes: plain operands, broadcasts from the LOW half of
    // a pair (op_sel_hi) and negations only -- never the (lo, hi) source selection of profiles/r05_pk_opsel_erratum.txt; the build's
    // assembly gate (tools/pk_opsel_fix.py --audit) and tests/test_erratum_gpu.py hold that line.
    // (Round 6 also tried a second loop body without the depth clamp for pose tiles whose depths all clear z_min -- three instructions
    // per pair less -- chosen per tile by a bound on |X|: no gain at C2 for the reason above, and 10-20 us more per launch at the
    // few-object shapes for the bound's extra pass and barriers; removed.)
    for (int t = 0; t < ntile; ++t) {
      const float* grp = ptab + (t * 4 + (col >> 2)) * 48 + (col & 3);
      const typename Proj::T ax = Proj::a(grp[kk * 4]), ay = Proj::a(grp[(4 + kk) * 4]), az = Proj::a(grp[(8 + kk) * 4]);
      const float4 aw4 = *reinterpret_cast<const float4*>(wtab + t * 16 + g4);
      // this lane's 4 poses: rows of K R (the back-projection), one float4 = one component of the four poses
      const float4* rows = reinterpret_cast<const float4*>(ptab + (t * 4 + kk) * 48);
      const float4 kxx = rows[0], kxy = rows[1], kxz = rows[2], kyx = rows[4], kyy = rows[5], kyz = rows[6], kzx = rows[8], kzy = rows[9],
                   kzz = rows[10];
      // Four resident tiles (two waves per SIMD, registers to spare): a software pipeline over the tiles -- tile i + 1's three
      // projections are issued in front of tile i's pair work, so that no wave waits out an MFMA's result latency in s_nops (29 -> 1
      // per pose tile).  With one or two tiles the compiler's own placement is the better one (the fence costs it its freedom).
      constexpr bool kPipe = (NPT == 4);
      floatx4 hxn = zero, hyn = zero, hzn = zero;
      if (kPipe) {
        hxn = Proj::mma(ax, rB[0], zero); hyn = Proj::mma(ay, rB[0], zero); hzn = Proj::mma(az, rB[0], zero);
      }
#pragma unroll
      for (int i = 0; i < NPT; ++i) {
        floatx4 hx, hy, hz;
        if (kPipe) {
          hx = hxn; hy = hyn; hz = hzn;
          if (i + 1 < NPT) {
            hxn = Proj::mma(ax, rB[i + 1], zero);
            hyn = Proj::mma(ay, rB[i + 1], zero);
            hzn = Proj::mma(az, rB[i + 1], zero);
            sched_fence();
          }
        } else {
          hx = Proj::mma(ax, rB[i], zero);
          hy = Proj::mma(ay, rB[i], zero);
          hz = Proj::mma(az, rB[i], zero);
        }
        const float4 w4 = rW[i];
        const f32x2 wu2 = {w4.x, w4.x}, cu2 = {w4.y, w4.y}, wv2 = {w4.z, w4.z}, cv2 = {w4.w, w4.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 hx2 = {hx[2 * h], hx[2 * h + 1]}, hy2 = {hy[2 * h], hy[2 * h + 1]};
          const float hz0 = hz[2 * h], hz1 = hz[2 * h + 1];
          const f32x2 aw2 = h ? f32x2{aw4.z, aw4.w} : f32x2{aw4.x, aw4.y};
          const f32x2 rz2 = {fast_rcp(clamp_below(hz0, zmin_v)), fast_rcp(clamp_below(hz1, zmin_v))};
          f32x2 rx2, ry2, ghx2, ghy2, ghz2;
          // Huber weight min(1, delta / rho) = min(1, rs) straight from the reciprocal norm (rho * rs = 1): ONE clamped multiply where
          // rho, min(rho, 1) and their product with rs took three.
          // d huber / d delta = max(rho - delta, 0) (/ delta) = coef |r|^2 - a min(|r|^2, 1) (residuals in units of delta; outlier:
          // a rho - a, inlier: a rho^2 - a rho^2 = 0): the first term is what A2x + A2y accumulate anyway, the second costs one clamped
          // multiply per pair and a packed fma (`gsat2`) where rho, rho (1 - c1) and the accumulation took three instructions per pair.
          f32x2 coef2, crx2, cry2;
          if (!BOUNDS) {
            // no projection clamp: the weight goes into the reciprocal depth once (w / z), which serves the residual and the
            // gradient w.r.t. (h_x, h_y) -- one multiply per pair less than projecting first
            const f32x2 wrx2 = rz2 * wu2, wry2 = rz2 * wv2;
            rx2 = fma2(hx2, wrx2, cu2);
            ry2 = fma2(hy2, wry2, cv2);
            const f32x2 s22 = fma2(rx2, rx2, fma2(ry2, ry2, tiny2));      // |r|^2 + 1e-30: one v_max less per pair than clamping
            const f32x2 c12 = {sat_mul(fast_rsqrt(s22[0]), one_v), sat_mul(fast_rsqrt(s22[1]), one_v)};
            coef2 = aw2 * c12;
            gsat2 = fma2(aw2, f32x2{sat_mul(s22[0], one_v), sat_mul(s22[1], one_v)}, gsat2);
            crx2 = coef2 * rx2;
            cry2 = coef2 * ry2;
            ghx2 = crx2 * wrx2;
            ghy2 = cry2 * wry2;
            ghz2 = -rz2 * fma2(ghx2, hx2, ghy2 * hy2);                      // = -(ghx p_x + ghy p_y)
          } else {
            const f32x2 ppx2 = hx2 * rz2, ppy2 = hy2 * rz2;                 // un-clamped projection
            const f32x2 px2 = {clamp_lu(ppx2[0], bd.lbx, bd.ubx), clamp_lu(ppx2[1], bd.lbx, bd.ubx)};
            const f32x2 py2 = {clamp_lu(ppy2[0], bd.lby, bd.uby), clamp_lu(ppy2[1], bd.lby, bd.uby)};
            rx2 = fma2(px2, wu2, cu2);
            ry2 = fma2(py2, wv2, cv2);
            const f32x2 s22 = fma2(rx2, rx2, fma2(ry2, ry2, tiny2));
            const f32x2 c12 = {sat_mul(fast_rsqrt(s22[0]), one_v), sat_mul(fast_rsqrt(s22[1]), one_v)};
            coef2 = aw2 * c12;
            gsat2 = fma2(aw2, f32x2{sat_mul(s22[0], one_v), sat_mul(s22[1], one_v)}, gsat2);
            crx2 = coef2 * rx2;
            cry2 = coef2 * ry2;
            f32x2 gpx2 = crx2 * wu2, gpy2 = cry2 * wv2;
            // the clamp passes no gradient where it is active: inside [lb, ub] <=> clamp(x) == x
            gpx2 = f32x2{(px2[0] == ppx2[0]) ? gpx2[0] : 0.f, (px2[1] == ppx2[1]) ? gpx2[1] : 0.f};
            gpy2 = f32x2{(py2[0] == ppy2[0]) ? gpy2[0] : 0.f, (py2[1] == ppy2[1]) ? gpy2[1] : 0.f};
            ghx2 = gpx2 * rz2;
            ghy2 = gpy2 * rz2;
            ghz2 = fma2(-ghx2, ppx2, -(ghy2 * ppy2));
          }
          // "in front of the depth clamp" as a 0/1 factor from ONE full-rate instruction per pair (front_scale above)
          ghz2 = ghz2 * f32x2{sat_fma(hz0, front_scale, front_off), sat_fma(hz1, front_scale, front_off)};
          // d/dw = crx * (px - u) = crx * rx / w and d/du = -crx * w: the per-point factors are applied once at the end
          A2x[i] = fma2(crx2, rx2, A2x[i]);
          A2y[i] = fma2(cry2, ry2, A2y[i]);
          A1x[i] = fma2(coef2, rx2, A1x[i]);
          A1y[i] = fma2(coef2, ry2, A1y[i]);
          const f32x2 kxx2 = h ? f32x2{kxx.z, kxx.w} : f32x2{kxx.x, kxx.y}, kyx2 = h ? f32x2{kyx.z, kyx.w} : f32x2{kyx.x, kyx.y},
                      kzx2 = h ? f32x2{kzx.z, kzx.w} : f32x2{kzx.x, kzx.y};
          const f32x2 kxy2 = h ? f32x2{kxy.z, kxy.w} : f32x2{kxy.x, kxy.y}, kyy2 = h ? f32x2{kyy.z, kyy.w} : f32x2{kyy.x, kyy.y},
                      kzy2 = h ? f32x2{kzy.z, kzy.w} : f32x2{kzy.x, kzy.y};
          const f32x2 kxz2 = h ? f32x2{kxz.z, kxz.w} : f32x2{kxz.x, kxz.y}, kyz2 = h ? f32x2{kyz.z, kyz.w} : f32x2{kyz.x, kyz.y},
                      kzz2 = h ? f32x2{kzz.z, kzz.w} : f32x2{kzz.x, kzz.y};
          gXv[i] = fma2(kxx2, ghx2, fma2(kyx2, ghy2, fma2(kzx2, ghz2, gXv[i])));
          gYv[i] = fma2(kxy2, ghx2, fma2(kyy2, ghy2, fma2(kzy2, ghz2, gYv[i])));
          gZv[i] = fma2(kxz2, ghx2, fma2(kyz2, ghy2, fma2(kzz2, ghz2, gZv[i])));
        }
      }
    }
    const float gsat = gsat2[0] + gsat2[1];
    {   // d/d delta of this chunk: sum_pairs coef |r|^2 (= the sums behind d/dw, before their per-point factors) - sum_pairs a min(|r|^2, 1)
      float a2 = 0.f;
#pragma unroll
      for (int i = 0; i < NPT; ++i) a2 += (A2x[i][0] + A2x[i][1]) + (A2y[i][0] + A2y[i][1]);
      gd += a2 - gsat;
    }
    // ---- outputs of this chunk: sums over the 4 pose groups of a point via MFMAs against indicator columns ----
    // The lane geometry is re-derived here behind an opaque copy of the lane index: as invariants of the chunk loop the four
    // indicator columns and the three 64-bit output addresses were hoisted in front of it and stayed live across the pose-tile
    // loop -- 10 VGPRs, which cost the bf16 instantiation 7 spilled dwords per lane (33 MB of scratch traffic per launch at C2).
    const int laneE = (int)f32_bits(to_vgpr(bits_f32((unsigned)lane)));
    const int colE = laneE & 15, g4E = (laneE >> 4) * 4;
    const float ind0 = (colE == 0) ? 1.f : 0.f, ind1 = (colE == 1) ? 1.f : 0.f, ind2 = (colE == 2) ? 1.f : 0.f,
                ind3 = (colE == 3) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      floatx4 D1 = zero;
      D1 = mfma_16x16x4((gXv[i][0] + gXv[i][1]) * hs.delta_sq, ind0, D1);
      D1 = mfma_16x16x4((gYv[i][0] + gYv[i][1]) * hs.delta_sq, ind1, D1);
      D1 = mfma_16x16x4((gZv[i][0] + gZv[i][1]) * hs.delta_sq, ind2, D1);
      const float4 w4 = rW[i];
      floatx4 D2 = zero;
      const float a1x = A1x[i][0] + A1x[i][1], a1y = A1y[i][0] + A1y[i][1], a2x = A2x[i][0] + A2x[i][1], a2y = A2y[i][0] + A2y[i][1];
      D2 = mfma_16x16x4(-w4.x * a1x * hs.delta_sq, ind0, D2);                               // d/du
      D2 = mfma_16x16x4(-w4.z * a1y * hs.delta_sq, ind1, D2);                               // d/dv
      D2 = mfma_16x16x4((w4.x != 0.f) ? a2x * hs.delta / w4.x : 0.f, ind2, D2);             // d/dwu
      D2 = mfma_16x16x4((w4.z != 0.f) ? a2y * hs.delta / w4.z : 0.f, ind3, D2);             // d/dwv
      const int nb = c0 + (wv + W * i) * 16 + g4E;    // D rows: points nb + r; column = lane & 15
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        if (n < p.N) {
          const size_t o = (size_t)b * p.N + n;
          if (colE < 3) gx3d[o * 3 + colE] = D1[r];
          if (colE < 2) gx2d[o * 2 + colE] = D2[r];
          else if (colE < 4) {
            if (park) gwl[n * 2 + (colE - 2)] = D2[r]; else gw2d[o * 2 + (colE - 2)] = D2[r];
          }
        }
      }
    }
  }
  PNP_PHASE(4);
  float one[1] = {gd * hs.delta};
  block_sum<1>(one, red);
  if (tid == 0) gdelta[v] = one[0];
  // delta = mean(w2d) * std * rel came from THIS w2d (Problem.delta_stats): its gradient reaches w2d as one number per
  // object, added here to what this workgroup has just written instead of by three launches of the caller's autograd.
  // (With the object split over workgroups the sum of the parts is not known here: delta_path_kernel, amis_kernels.hip.)
  if (fold) {
    const float add = (one[0] * p.delta_stats[(size_t)b * 4 + 1]) * (p.delta_relative / (2.0f * (float)p.N));
    __syncthreads();        // this workgroup's gw2d values (LDS or global) are visible to all of its threads behind the barrier
    float* row = gw2d + (size_t)b * p.N * 2;
    if (park) {
      for (int i = tid; i < 2 * p.N; i += T) row[i] = gwl[i] + add;       // the only pass over grad_w2d, coalesced
    } else {                // (no LDS left for the rows: read back what was just written)
      for (int i = tid; i < 2 * p.N; i += T) row[i] += add;
    }
  }
  PNP_PHASE(5);
  PNP_PHASES_FLUSH(6);
}

// per-phase cycle totals of this file's kernel (tuning builds; -1 otherwise)
int tuning_bwd_phase_cycles(unsigned long long* out, int reset) { return tuning::read_cycles(out, reset, false); }

template <class F>
static int dispatch_bwd_npt(int npt, F&& f) {
  switch (npt) {
    case 1: return f(ic<1>{});
    case 2: return f(ic<2>{});
    default: return f(ic<4>{});
  }
}

// returns 1 when the shape is not supported (caller falls back to amis_backward_kernel)
int launch_amis_backward_mfma(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                              int mc_samples, const float* pose_init, const float* grad_cost_init, float* grad_x3d,
                              float* grad_x2d, float* grad_w2d, float* grad_delta, int nsplit, hipStream_t st) {
  const Problem d = to_device_problem(prob);
  const int P = mc_samples + ((pose_init && grad_cost_init) ? 1 : 0);
  const int P16 = ((P + 15) / 16) * 16 + 16;
  size_t smem = sizeof(float) * (15 * (size_t)P16 + 80 + kDropHistFloats);
  if (smem > 160 * 1024) return 1;
  // grad_w2d rows parked in LDS while the threshold's gradient is folded in (kernel comment): when the fold applies and the
  // rows fit next to three workgroups' pose tables per CU
  int gw_rows = 0;
  if (prob->delta_stats != nullptr && nsplit == 1 && 3 * (smem + sizeof(float) * 2 * (size_t)d.N) <= 160 * 1024) {
    gw_rows = d.N;
    smem += sizeof(float) * 2 * (size_t)d.N;
  }
  // 4 waves x NPT <= 4 point tiles of 16 per chunk; larger N loops over chunks of 256 points against the LDS-resident pose table.
  // Measured at C2 in round 1: 4x4 (2 chunks) 1.07 ms, 4x8 1.11, 8x4 1.22.
  // Few objects (fewer than two waves per SIMD otherwise): 8 waves, one chunk of 512 points (B = 32 / 256: -7..9 %).
  const int ptiles = (d.N + 15) / 16;
  int waves = (d.B < 512 && ptiles > 16) ? 8 : 4, npt = 1;
  while (npt < 4 && waves * npt < ptiles) npt *= 2;
  // Round 6, packed pair loop with pair accumulators (same-box A/Bs, profiles/r06_bwd_packed.txt): without a projection clamp 4 x 4
  // wins -- 784 ... 819 us at 206-248 VGPRs (two waves per SIMD) against 808 ... 838 us for 4 x 2 at 152 VGPRs (three): with two point-poses
  // per instruction the fewer instructions per pair of the four-tile loop (22.6 against 24.0) weigh more than the third wave.  With the
  // clamp the four-tile loop needs 218 VGPRs and two tiles win where the grid fills the chip: 913 against 945 us.  (The scalar loop of
  // round 5 preferred two tiles either way: 905 at 4 x 4, 868 at 4 x 2 with four waves.)  EPROPNP_TUNE=bwd_mfma=<waves>,<tiles> overrides.
  if (has_bounds(prob) && waves == 4 && npt == 4 && d.B >= 2 * device_cu_count() && 3 * smem <= 160 * 1024) npt = 2;
  if (nsplit > 1) {      // 4 waves x the fewest tiles that still cover N with nsplit chunks in flight
    waves = 4; npt = 1;
    while (npt < 4 && waves * npt * nsplit < ptiles) npt *= 2;
  }
  { int ov[2]; if (tune_ints("bwd_mfma", ov, 2) && ov[0] >= 1 && ov[0] <= 8 && (ov[1] == 1 || ov[1] == 2 || ov[1] == 4)) { waves = ov[0]; npt = ov[1]; } }
  // Projection on the bf16 matrix path (kernel comment) wherever the split operands fit the register budget of three waves per
  // SIMD (<= 4 resident point tiles: 4 VGPRs per tile instead of 1).  C2 backward 0.954 -> 0.899 ms, bounded 1.111 -> 1.037 ms,
  // Det shape neutral (profiles/r04_bwd_bf16_projection.txt).  EPROPNP_BWD_PROJ=f32 | bf16 forces either.
  // (With a projection clamp and four resident tiles the bf16 instantiation is a two-waves-per-SIMD kernel, see above.  It is taken
  // whatever the grid: which projection arithmetic an object gets must not depend on how many objects share the launch or on how
  // its points are split over workgroups -- the per-point gradients of the split and the unsplit launch are the same bits.)
  const dim3 grid(padded_object_grid(d.B * nsplit)), block(64 * waves);
  bool bf16 = true;
  if (const char* e = getenv("EPROPNP_BWD_PROJ")) bf16 = (e[0] == 'f') ? false : (e[0] == 'b' ? true : bf16);
  dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
    auto launch = [&](auto kern) -> int {
      allow_dynamic_lds((const void*)kern, smem);
      PNP_LAUNCH(kern, grid, block, smem, st, d, pose_samples, grad_logweights, mc_samples, pose_init, grad_cost_init, P16,
                 grad_x3d, grad_x2d, grad_w2d, grad_delta, nsplit, backward_drop_eps(), gw_rows);
      return 0;
    };
    if (bf16)
      return (npt == 1)   ? launch(amis_backward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, 1, true>)
             : (npt == 2) ? launch(amis_backward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, 2, true>)
                          : launch(amis_backward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, 4, true>);
    return dispatch_bwd_npt(npt, [&](auto NPT) -> int {
      return launch(amis_backward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, decltype(NPT)::value>);
    });
  });
  return check_launch("amis_backward_mfma_kernel");
}

}  // namespace pnp
