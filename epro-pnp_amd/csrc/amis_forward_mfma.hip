// amis_forward_mfma.hip -- AMIS forward with the pose x point projection on the matrix cores.
//
// Same algorithm and LDS-resident sampler state as amis_forward_kernel (amis_kernels.hip); only the cost sweep
// differs.  The projection h = (K R | K t) (X,Y,Z,1)^T of 16 poses x 16 points is ONE matrix instruction per image row
// (x, y, z):
//   * register mode (an object's points resident in the waves' registers: the default up to 2048 points per part), BF16:
//     v_mfma_f32_16x16x32_bf16 on operands split into three bf16 pieces each (wave_ops.h: ProjOp<true>; fp32-level accuracy,
//     7.8 ns per instruction and SIMD).  The pose rows are split on the fly, the point tiles once; a split operand is a 4-VGPR
//     tuple per image row, so the weights stay out of the B operands (two more multiplies per point-pose);
//   * otherwise v_mfma_f32_16x16x4_f32 (exact f32, a k-ordered fmaf chain; 14.3 ns): the LDS-streamed mode, and register mode
//     under EPROPNP_FWD_PROJ=f32 -- there, without a projection clamp, the weights are folded into the B operands of the x and
//     y rows (5 VGPRs per resident tile).  The fp32 MFMA has the VALU's FMA rate and does not overlap with it
//     (tools/ubench/mfma_valu_overlap.hip): what it buys is issue slots and registers, not a second pipe.
// What stays on the VALU is the perspective divide, the weighted residual and the Huber kernel (2 transcendentals + ~10
// simple ops per point-pose).
//   A operand (16 poses x k): lane l holds row[pose l&15][k-slice l>>4]   <- LDS pose table, x | y | z rows of (K R | K t)
//   B operand (k x 16 points): lane l holds (X,Y,Z,1)[l>>4] of point l&15  <- registers (register mode) / LDS point table
//   D (16 x 16): lane l holds poses 4*(l>>4)+r, r = 0..3, at point l&15   -> 4 point-poses per lane per tile
// Register mode: the workgroup's waves split the POINTS and sweep every pose tile; a pose's cost is the row sum over the 16
// lanes that hold its 16 points, the waves' partial costs meet in LDS.  LDS-streamed mode (any N): waves split the pose tiles,
// chunks of <= kChunk points stream through LDS.
#include "amis_common.h"
#include "dispatch.h"

namespace pnp {

// (tuning builds count shader-clock cycles per phase -- initial fit + load | draw | sweep | weights | refit | store --: PNP_PHASE,
// tuning.h; read back through epropnp_tuning_phase_cycles)

// Two point-poses at a time, written on 2-vectors (wave_ops.h: f32x2, fma2) so that the multiplies / FMAs become
// v_pk_mul_f32 / v_pk_fma_f32.  Packed ops run at the scalar flop rate on gfx950, but the transcendental ops between them
// (rcp, sqrt) then cost ~3 ns instead of ~5.6 ns per wave (tools/ubench: "2 trans : 6 pk_fma" vs "2 trans : 6 fma") -- a
// fifth of this loop.

// Huber costs of the 4 poses a lane holds for one point (MFMA outputs hx, hy, hz) added to acc2 = {poses 0,1}, {2,3}
// The weights in w4 (and those folded into hx, hy) are pre-divided by the object's Huber threshold delta, so the
// residual norm rho is in units of delta, huber / delta^2 = m (rho - m / 2) with m = min(rho, 1), and the caller scales
// the pose's sum by delta^2 once.
// FOLDED: hx, hy are the MFMA results against the point's B operand pre-scaled by wu / wv (register mode without a
// projection clamp), so the residual is ONE fma per coordinate: r = (wu h_x) / z - u wu.
template <bool BOUNDS, bool FOLDED = false>
__device__ __forceinline__ void huber_cost_4(const floatx4& hx, const floatx4& hy, const floatx4& hz, const float4& w4,
                                             float zmin_v, float one_v, const Bounds& bd, f32x2 (&acc2)[2]) {
  static_assert(!(BOUNDS && FOLDED), "the clamp acts on the un-weighted projection");
  const f32x2 wu2 = {w4.x, w4.x}, wv2 = {w4.y, w4.y}, cu2 = {w4.z, w4.z}, cv2 = {w4.w, w4.w};
  const f32x2 mhalf = {-0.5f, -0.5f};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // z = max(h_z, z_min) as one instruction (compare+select measured 3.4 ns per pair against 1.8 ns, tools/ubench)
    const float z0 = clamp_below(hz[2 * h], zmin_v), z1 = clamp_below(hz[2 * h + 1], zmin_v);
    const f32x2 rz2 = {fast_rcp(z0), fast_rcp(z1)};
    const f32x2 hx2 = {hx[2 * h], hx[2 * h + 1]}, hy2 = {hy[2 * h], hy[2 * h + 1]};
    f32x2 rx2, ry2;
    if (FOLDED) {
      rx2 = fma2(hx2, rz2, cu2);
      ry2 = fma2(hy2, rz2, cv2);
    } else {
      f32x2 px2 = hx2 * rz2, py2 = hy2 * rz2;
      if (BOUNDS) {
        px2 = f32x2{clamp_lu(px2[0], bd.lbx, bd.ubx), clamp_lu(px2[1], bd.lbx, bd.ubx)};
        py2 = f32x2{clamp_lu(py2[0], bd.lby, bd.uby), clamp_lu(py2[1], bd.lby, bd.uby)};
      }
      rx2 = fma2(px2, wu2, cu2);
      ry2 = fma2(py2, wv2, cv2);
    }
    const f32x2 s2 = fma2(rx2, rx2, ry2 * ry2);
    const f32x2 rho2 = {fast_sqrt(s2[0]), fast_sqrt(s2[1])};
    const f32x2 m2 = {sat_mul(rho2[0], one_v), sat_mul(rho2[1], one_v)};
    acc2[h] = fma2(m2, fma2(mhalf, m2, rho2), acc2[h]);       // huber / delta^2 = m (rho - m / 2), m = min(rho, 1)
  }
}

constexpr int kChunk = 1024;   // points per LDS chunk (32 KiB of point tables)

struct MfmaShape {
  int chunk;     // points per chunk, multiple of 16
  int s16;       // rows of the LDS pose table: the samples of an iteration rounded up to 16 -- or, in the SPILL variant, of
                 // one TILE of an iteration's samples when a whole iteration does not fit LDS (draw -> sweep per tile)
  int ahead;     // 1: LDS holds the [s][8] buffer for base noise drawn ahead of the proposal fit (0: drawn inline)
  int chunks;    // register mode, > 1: the object's point tiles go through the waves' registers in `chunks` groups per iteration
};

// NPT > 0: the workgroup's waves split the POINTS, each wave keeps its NPT point tiles (B operand + residual
// constants, 5 VGPRs per tile) in registers for the whole kernel and sweeps every pose tile; per-wave partial costs
// meet in LDS.  NPT == 0: waves split the pose tiles and points stream through LDS in chunks (any N).
// SPILL (NPT == 0 only): the per-sample sampler state -- samples, costs, mixture densities, log-weights, (pose_len + 3) S
// floats per object -- lives in a global scratch buffer instead of LDS, for mc_samples beyond what 160 KiB hold
// (the reference has no such limit).  Same code: the arrays are reached through pointers either way, every hand-over
// between threads goes through a workgroup barrier, and a workgroup's global accesses share one L1.
// BF16 (register mode only): the projection on v_mfma_f32_16x16x32_bf16 with bf16x3-split operands (wave_ops.h: ProjOp) -- 55 % of the
// fp32 MFMA's time at fp32-level accuracy.  A split B operand takes 4 VGPRs per resident point tile and a register tuple per image
// row cannot double as the weighted operand, so the weights are NOT folded into the B operands here (two more multiplies per
// point-pose) and the 6-DoF kernel is compiled for three waves per SIMD instead of four (152 VGPRs): still -9 % at C2
// (profiles/r04_fwd_bf16_projection.txt).
template <int DOF, bool BOUNDS, int NPT, bool SPILL = false, bool SPLIT = false, bool BF16 = false, bool CHUNKED = false>
// (SPLIT grids are sized for one workgroup per CU: two waves per SIMD -- 256 VGPRs -- leave room for a second such launch and
// for the part-recomputation path's second copy of the sweep without spilling)
__global__ __launch_bounds__(512, SPLIT ? 2 : (DOF == 6 ? (NPT <= 8 ? (BF16 ? PNP_FWD_BF16_MINW : PNP_FWD_MINW) : 2) : (NPT <= 2 ? 3 : 2))) void amis_forward_mfma_kernel(Problem p, AmisParams a_in, MfmaShape sh,
                                                                  const float* __restrict__ pose_opt,
                                                                  const float* __restrict__ pose_cov,
                                                                  const float* __restrict__ noise,
                                                                  float* __restrict__ pose_samples,
                                                                  float* __restrict__ logweights,
                                                                  float* __restrict__ proposals,
                                                                  float* __restrict__ spill, int nsplit,
                                                                  float* __restrict__ xch) {
  static_assert(!SPILL || NPT == 0, "the spill variant streams the points");
  static_assert(!SPLIT || NPT > 0, "the split over workgroups is a register-mode variant");
  constexpr int PL = PoseLen<DOF>::value;
  // nsplit = G > 1 (few objects, register mode): G workgroups share one object.  Each runs the whole sampler -- draws, weights
  // and proposal fits are deterministic, so the G copies stay identical -- but sweeps only every G-th group of point tiles;
  // the partial costs of an iteration meet in global memory behind a per-object arrival counter (below).  All parts of an
  // object sit on the same XCD (workgroup g -> XCD g % 8), so the exchange stays in one L2.
  // (a template parameter: the one-workgroup-per-object instantiations carry none of this -- the scalar registers it keeps
  // live spilled two VGPRs of the C2 kernel otherwise)
  const int G = SPLIT ? nsplit : 1;
  static_assert(!CHUNKED || (NPT > 0 && !SPLIT && !SPILL), "chunks are a register-mode variant of the one-workgroup kernel");
  // chunked register mode (launcher comment); a template parameter: the loop around load_tiles / sweep_regs costs the other
  // instantiations their register allocation (C2 kernel: 83 instead of 7 spilled VGPRs with a run-time chunk count)
  const int CH = CHUNKED ? sh.chunks : 1;
  const int GT = SPLIT ? G : CH;                  // groups the point tiles are dealt to: workgroups of the object, or chunks in time
  int b, part = 0;
  if (SPLIT) {
    const int g = (int)blockIdx.x, per = (p.B + 7) >> 3, idx = g >> 3;
    part = idx % G;
    b = (g & 7) * per + idx / G;
  } else {
    b = object_of_block(p.B);
  }
  if (b >= p.B) return;
  AmisParams a = a_in;
  if (a.offset_dev != nullptr) a.offset += *a.offset_dev;
  const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = lane_id(), wv = wave_id(), W = T >> 6;
  const int S = a.S, K = a.K, s = S / K, s16 = sh.s16, NC = sh.chunk;

  PNP_PHASES_BEGIN(6);
  PNP_DYN_SMEM(float, smem);
  constexpr bool kRegs = NPT > 0;
  static_assert(!BF16 || NPT > 0, "the split projection is a register-mode variant");
  constexpr bool kBf = BF16;
  typedef ProjOp<kBf> Proj;
  constexpr bool kFold = kRegs && !BOUNDS && !kBf;
  const int WPs = kRegs ? (G > 1 ? G : W) : 1;      // point slices whose partial costs are summed in amis_weights
  // split: W per-wave rows of the part in the registers + G gathered rows (one per part) + the missing-parts word
  const int cpart_rows = kRegs ? (SPLIT ? W + G : W) : 1;
  float* ptab = smem;                 // [s16][12]   x | y | z rows of (K R | K t)          (16-B aligned)
  float* pB = ptab + 12 * s16;        // [NC][4]     (X, Y, Z, 1)        (NC = 0 in register mode)
  float* pW = pB + 4 * NC;            // [NC][4]     (wu, wv, -u wu, -v wv)
  float* smp = SPILL ? spill + (size_t)b * (PL + 3) * S : pW + 4 * NC;       // [PL][S]
  float* cst = smp + PL * S;          // [S]
  float* mixl = cst + S;              // [S]
  float* lgw = mixl + S;              // [S]
  // (the sampler arrays hold (PL + 3) S floats -- 10 S or 7 S, a multiple of 4 only for even S (6-DoF) / S % 4 == 0 (4-DoF); what
  // follows them is read as float4 (cpart rows, the refit's rred): the block is rounded up, here and in the launcher's lds_bytes)
  float* cpart = SPILL ? pW + 4 * NC : smp + (((PL + 3) * S + 3) & ~3);      // [WPs][s16]
  float* gath = cpart + W * s16;             // [G][s16] (SPLIT) the parts' partial costs of the current iteration
  float* prop = cpart + cpart_rows * s16 + (SPLIT ? 4 : 0);    // [K][kPropStride]  (SPLIT: + the missing-parts word)
  float* red = prop + K * kPropStride;   // [256]
  float* rred = red + 256;               // [kRefitRedFloats] the refit's transposed reductions (amis_common.h: wave_sum_t)
  float* nzb = (s <= T) ? ptab : rred + kRefitRedFloats;   // [s][8] base noise drawn ahead; shares the pose table's LDS when
                                              // one sample per lane suffices (amis_draw separates the two uses)

  float Kc[9], delta;
  Bounds bd;
  load_camera<BOUNDS>(p, b, Kc, bd, delta);
  const float zmin_v = to_vgpr(p.z_min), one_v = to_vgpr(1.0f);
  // residuals in units of delta (huber_cost_4, huber_scale)
  const HuberScale hs = huber_scale(delta);
  const float inv_delta = hs.inv_delta, delta_sq = hs.delta_sq;

  const bool tiled = SPILL && s > s16;          // an iteration's samples go through the pose table in tiles of s16
  AmisCtx cx;
  cx.ptab = ptab; cx.smp = smp; cx.cst = cst; cx.mixl = mixl; cx.lgw = lgw; cx.cpart = SPLIT ? gath : cpart; cx.prop = prop; cx.red = red;
  cx.S = S; cx.K = K; cx.s = s; cx.T = T; cx.tid = tid; cx.b = b; cx.cstride = s16;
  cx.nzb = (noise == nullptr && W > 1 && sh.ahead) ? nzb : nullptr;
  cx.rred = rred;

  if (!tiled)
    for (int i = tid; i < 12 * (s16 - s); i += T) ptab[12 * s + i] = 0.f;   // padding poses of the last pose tile
  if (tid < (DOF == 6 ? 2 : 1))      // 6-DoF: lane 1 fits the translation factor inside lane 0's rotation fit
    initial_fit<DOF>(pose_opt + (size_t)b * PL, pose_cov + (size_t)b * DOF * DOF, a.eps, a.dispersion, prop, tid);
  if (part == 0 && tid == T - 1) denormalise_pose_opt<DOF>(a, pose_opt, b);      // (a lane of the last wave: wave 0 is busy with the fit)
  const int nchunk = kRegs ? 1 : (p.N + NC - 1) / NC;
  auto load_chunk = [&](int c0) {
    const int cnt = min(NC, ((p.N - c0 + 15) >> 4) << 4);
    for (int n = tid; n < cnt; n += T) {
      const Point q = load_point(p, b, c0 + n);          // zero weight beyond N
      reinterpret_cast<float4*>(pB)[n] = make_float4(q.X, q.Y, q.Z, 1.0f);
      const float wu = q.wu * inv_delta, wv = q.wv * inv_delta;
      reinterpret_cast<float4*>(pW)[n] = make_float4(wu, wv, -q.u * wu, -q.v * wv);
    }
  };
  // register mode: this wave's point tiles q = wv + W * i of part `pt`, lane = (point column, k)
  typename Proj::T rB[kRegs ? NPT : 1];
  float4 rW[kRegs ? NPT : 1];
  auto load_tiles = [&](int pt) {
#pragma unroll
    for (int i = 0; i < (kRegs ? NPT : 1); ++i) {
      const Point q = load_point(p, b, ((pt * W + wv) + GT * W * i) * 16 + (lane & 15));      // zero weight beyond N
      const int k4 = lane >> 4;
      const float bval = (k4 == 0) ? q.X : (k4 == 1) ? q.Y : (k4 == 2) ? q.Z : 1.0f;
      rB[i] = Proj::b(bval);
      // without a projection clamp the weights are folded into the B operands of the x and y rows (same 5 VGPRs)
      const float wu = q.wu * inv_delta, wv = q.wv * inv_delta;
      rW[i] = kFold ? make_float4(bval * wu, bval * wv, -q.u * wu, -q.v * wv) : make_float4(wu, wv, -q.u * wu, -q.v * wv);
    }
  };
  if (kRegs) {
    if constexpr (!CHUNKED) load_tiles(part);
  } else if (nchunk == 1) {
    load_chunk(0);
  }
  amis_base_noise<DOF>(cx, a, 0, 64);     // waves 1.. draw the first iteration's base noise while lane 0 fits proposal 0
  __syncthreads();

  const int g4 = (lane >> 4) * 4, col = lane & 15, kk = lane >> 4;
  // register mode: every pose tile of the iteration against the point tiles in this wave's registers -> cpart[wv][pose]
  // (issuing the next POSE tile's first MFMAs ahead of the current tile's VALU work, to fill the wait states at the
  // head of this loop, was measured in round 1: 0.655 vs 0.641 ms at C2 -- the hand-off copies cost more than the s_nops.
  // Round 6 pipelines over the resident POINT tiles inside a pose tile instead -- tile i + 1's three projections in front of
  // tile i's Huber sweep, pinned by a scheduling fence: 48 -> 22 s_nops per pose tile, no register more at eight tiles,
  // 539 -> 514 us at C2, 7.22 -> 6.77 ms at the C5 shard; same bits.  PNP_FWD_PIPE, tuning.h)
  auto sweep_regs = [&](bool accumulate = false) {      // accumulate: a later chunk of the same iteration adds to the wave's row
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < (s16 >> 4); ++t) {
      const float* arow = ptab + 12 * (t * 16 + col) + kk;
      const typename Proj::T ax = Proj::a(arow[0]), ay = Proj::a(arow[4]), az = Proj::a(arow[8]);
      f32x2 acc2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
      constexpr int NT = kRegs ? NPT : 1;
      constexpr bool kPipe = (PNP_FWD_PIPE != 0) && !kFold && NT >= 4;      // (tuning.h: software pipeline over the resident tiles)
      floatx4 hxn = zero, hyn = zero, hzn = zero;
      if (kPipe) {
        hxn = Proj::mma(ax, rB[0], zero); hyn = Proj::mma(ay, rB[0], zero); hzn = Proj::mma(az, rB[0], zero);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        floatx4 hx, hy, hz;
        if constexpr (kFold) {
          hx = mfma_16x16x4(ax, rW[i].x, zero);
          hy = mfma_16x16x4(ay, rW[i].y, zero);
          hz = mfma_16x16x4(az, rB[i], zero);
        } else if constexpr (kPipe) {
          hx = hxn; hy = hyn; hz = hzn;
          if (i + 1 < NT) {
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
        huber_cost_4<BOUNDS, kFold>(hx, hy, hz, rW[i], zmin_v, one_v, bd, acc2);
      }
      // the four poses' sums over this row's 16 points: lane col < 4 ends up with pose g4 + col (wave_ops.h: row_sum16_of4)
      const float acc[4] = {acc2[0][0], acc2[0][1], acc2[1][0], acc2[1][1]};
      const float tot = row_sum16_of4(acc);
      if (col < 4) {
        float* dst = cpart + wv * s16 + t * 16 + g4 + col;
        if constexpr (CHUNKED) *dst = accumulate ? fmaf(tot, delta_sq, *dst) : tot * delta_sq;
        else *dst = tot * delta_sq;
      }
    }
  };
  PNP_PHASE(0);
  unsigned gone_parts = 0u;     // (SPLIT) sibling parts that have timed out once
  for (int it = 0; it < K; ++it) {
    for (int n0 = 0; n0 < (SPILL ? s : 1); n0 += (SPILL ? s16 : 1)) {        // (exactly one pass unless `tiled`)
    const int cnt = min(s16, s - n0), cnt16 = (cnt + 15) & ~15;
    if (tiled) {
      if (n0 > 0) __syncthreads();              // the previous tile's costs have left cpart, its poses are done with
      for (int i = tid; i < 12 * (cnt16 - cnt); i += T) ptab[12 * cnt + i] = 0.f;
    }
    if (!PNP_ABLATED(a, 32) || it == 0)   // (tuning, bit5: the sweep re-uses the first iteration's pose table)
      amis_draw<DOF>(cx, p, a, it, Kc, noise, part == 0 ? pose_samples : nullptr, n0, tiled ? cnt : -1);
    __syncthreads();
    PNP_PHASE(1);

    // ---------------- cost sweep: 16 x 16 (pose, point) tiles on the matrix pipe ----------------
    if (PNP_ABLATED(a, 1)) {
      for (int n = tid; n < s; n += T) cpart[n] = 1.0f;
    } else if (kRegs) {
      if constexpr (!CHUNKED) {
        sweep_regs();
      } else {        // the point tiles in CH groups through the registers: load, split, sweep every pose tile, next group
        for (int c = 0; c < CH; ++c) {
          load_tiles(c);
          sweep_regs(c > 0);
        }
      }
    } else
    for (int ch = 0; ch < nchunk; ++ch) {
      const int c0 = ch * NC;
      if (nchunk > 1) {
        if (ch > 0 || it > 0) __syncthreads();    // everyone is done with the previous chunk
        load_chunk(c0);
        __syncthreads();
      }
      const int npt = (min(NC, p.N - c0) + 15) >> 4;
      for (int t = wv; t < (cnt16 >> 4); t += W) {
        const float* arow = ptab + 12 * (t * 16 + col) + kk;
        const float ax = arow[0], ay = arow[4], az = arow[8];
        f32x2 acc2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
        // software pipeline: tile q+1's operands are fetched and its three MFMAs issued before tile q's VALU work,
        // so the matrix pipe (32 cycles per MFMA) runs underneath the VALU instructions of the previous tile
        float bq = pB[4 * col + kk];
        float4 w4 = reinterpret_cast<const float4*>(pW)[col];
        floatx4 hx = mfma_16x16x4(ax, bq, zero), hy = mfma_16x16x4(ay, bq, zero), hz = mfma_16x16x4(az, bq, zero);
        for (int q = 0; q < npt; ++q) {
          const int qn = min(q + 1, npt - 1);
          const float bn = pB[4 * (qn * 16 + col) + kk];
          const float4 wn = reinterpret_cast<const float4*>(pW)[qn * 16 + col];
          const floatx4 hxn = mfma_16x16x4(ax, bn, zero);
          const floatx4 hyn = mfma_16x16x4(ay, bn, zero);
          const floatx4 hzn = mfma_16x16x4(az, bn, zero);
          huber_cost_4<BOUNDS>(hx, hy, hz, w4, zmin_v, one_v, bd, acc2);
          hx = hxn; hy = hyn; hz = hzn; w4 = wn;
        }
        const float acc[4] = {acc2[0][0], acc2[0][1], acc2[1][0], acc2[1][1]};
        const float tot = row_sum16_of4(acc);
        if (col < 4) {
          const int pose = t * 16 + g4 + col;
          cpart[pose] = (ch == 0) ? tot * delta_sq : fmaf(tot, delta_sq, cpart[pose]);
        }
      }
    }
    __syncthreads();
    if (tiled)      // this tile's costs -> the (global) cost array; amis_weights is told they are there
      for (int m = tid; m < cnt; m += T) cst[it * s + n0 + m] = cpart[m];
    }             // tiles of the iteration
    if (kRegs && SPLIT) {
      // ---- exchange of the partial costs between the G parts of this object (wave_ops.h: xwg_*) ----
      // own row = sum over this workgroup's waves (fixed order) -> global slot [b][it][part][s16] and gath[part]; every
      // thread then polls the words it gathers from the other parts -- one store and one load round trip, no counter, one
      // barrier.  A part that does not show up within the timeout (not resident: CU mask, partitioned GPU, a foreign kernel
      // holding CUs) is RECOMPUTED here: its point tiles are loaded into this workgroup's registers and swept with the same
      // lanes in the same order, i.e. to the same bits, so the outputs never depend on co-residency, only the time does.
      unsigned* slot = reinterpret_cast<unsigned*>(xch) + (((size_t)b * K + it) * G) * s16;
      unsigned* missw = reinterpret_cast<unsigned*>(gath + G * s16);
      for (int m = tid; m < s; m += T) {
        float c = cpart[m];
        for (int w = 1; w < W; ++w) c += cpart[w * s16 + m];
        const unsigned bits = xwg_payload(c);
        xwg_store(slot + part * s16 + m, bits);
        gath[part * s16 + m] = bits_f32(bits);
      }
      if (tid == 0) *missw = 0u;
      __syncthreads();
      unsigned miss = 0u;      // parts that timed out in an earlier iteration, or for an earlier slot of this thread, are not waited
      for (int i = tid; i < G * s; i += T) {      // for again: what is there is taken, the rest recomputed below
        const int q = i / s, m = i - q * s;
        if (q == part) continue;
        const unsigned v = xwg_poll(slot + q * s16 + m, (((miss | gone_parts) >> q) & 1u) ? 0u : a.split_timeout);
        if (v == kXwgEmpty) miss |= 1u << q; else gath[q * s16 + m] = bits_f32(v);
      }
      if (miss) atomicOr(reinterpret_cast<int*>(missw), (int)miss);
      __syncthreads();
      unsigned todo = *missw;                   // the same in every thread
      gone_parts |= todo;
      if (todo) {
        if (tid == 0) raise_status(p, EPROPNP_ST_SPLIT_TIMEOUT, b);        // informational: this launch ran slower, not wrong
        for (int q = 0; q < G; ++q) {
          if (!((todo >> q) & 1u)) continue;
          load_tiles(q);
          sweep_regs();
          __syncthreads();
          for (int m = tid; m < s; m += T) {
            float c = cpart[m];
            for (int w = 1; w < W; ++w) c += cpart[w * s16 + m];
            gath[q * s16 + m] = bits_f32(xwg_payload(c));
          }
          __syncthreads();
        }
        load_tiles(part);
      }
    }
    PNP_PHASE(2);

    if (tiled) __syncthreads();
    // The phases behind the sweep address LDS by thread index (cst / mixl / lgw / cpart + tid + k s ...).  As invariants of the iteration
    // loop those addresses were computed once in front of it and stayed live across the sweep -- six to eight VGPRs the eight-tile
    // instantiation does not have: they were spilled before every sweep and reloaded behind it (28-32 B per lane, 30 MB of scratch
    // traffic per launch at C2).  Behind an opaque copy of the thread index they are recomputed here, per iteration, for a few integer adds.
    AmisCtx cxw = cx;
    cxw.tid = (int)f32_bits(to_vgpr(bits_f32((unsigned)tid)));
    amis_weights<DOF>(cxw, a, it, tiled ? 0 : WPs);
    __syncthreads();
    PNP_PHASE(3);
    if (it == K - 1) break;
    amis_refit<DOF>(cxw, a, it);
    PNP_PHASE(4);
  }

  if (part == 0) {   // numerical events for the caller's status word (include/epropnp_hip.h); no-op without one
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
  if (proposals != nullptr && part == 0)
    for (int i = tid; i < K * kPropStride; i += T) proposals[(size_t)b * K * kPropStride + i] = prop[i];
  PNP_PHASE(5);
  advance_counters(a, p.B * G);       // (every workgroup that passed the `b >= p.B` test above: B objects x G parts)
  PNP_PHASES_FLUSH(6);
}

// per-phase cycle totals of this file's kernel (tuning builds; -1 otherwise)
int tuning_phase_cycles(unsigned long long* out, int reset) { return tuning::read_cycles(out, reset, true); }

template <class F>
static int dispatch_npt(int npt, F&& f) {
  switch (npt) {
    case 0: return f(ic<0>{});
    case 1: return f(ic<1>{});
    case 2: return f(ic<2>{});
    case 4: return f(ic<4>{});
    case 8: return f(ic<8>{});
    case 12: return f(ic<12>{});
    default: return f(ic<16>{});
  }
}

// Parts per object for the split over workgroups: the most parts (<= 8) that leave every wave two point tiles and keep the
// whole grid at ONE workgroup per CU of THIS device (hipDeviceProp.multiProcessorCount: a partitioned MI355X reports its own
// count) -- the parts of an object exchange partial costs every iteration, which is fast only while they all run at once
// (each CU holds at least two of these workgroups, which leaves room for a second such launch on another stream).  Results
// never depend on that: a part that is not there in time is recomputed by its siblings (kernel comment).
// Two parts do not pay for the exchange.  EPROPNP_FWD_SPLIT=<G> overrides (1: off).
static int forward_split_parts(int B, int ptiles) {
  const long wgs = padded_object_grid(B), cus = device_cu_count();
  int g = 8;
  while (g > 1 && (wgs * g > cus || 8 * g > ptiles)) g >>= 1;
  if (g < 4) g = 1;
  { int ov[1]; if (env_ints("EPROPNP_FWD_SPLIT", ov, 1) && (ov[0] == 1 || ov[0] == 2 || ov[0] == 4 || ov[0] == 8) && wgs * ov[0] <= 4096 && 4 * ov[0] <= ptiles) g = ov[0]; }
  return g;
}

unsigned long long amis_forward_split_bytes(const epropnp_problem* prob, int mc_samples, int num_iter) {
  if (prob == nullptr || prob->num_obj <= 0 || num_iter <= 0 || mc_samples % num_iter != 0) return 0;
  const int ptiles = (prob->num_pts + 15) / 16;
  const int g = forward_split_parts(prob->num_obj, ptiles);
  if (g <= 1 || (ptiles + 4 * g - 1) / (4 * g) > 8) return 0;      // a part must fit 4 waves x 8 register-resident tiles
  const int s16 = ((mc_samples / num_iter + 15) / 16) * 16;
  return sizeof(float) * (unsigned long long)prob->num_obj * num_iter * g * s16;
}

int launch_amis_forward_mfma(const epropnp_problem* prob, const epropnp_amis_params* am, const float* pose_opt,
                             const float* pose_cov, const float* noise, float* pose_samples, float* logweights,
                             float* proposals, hipStream_t st, const DenormOut* dn) {
  const Problem d = to_device_problem(prob);
  const int S = am->mc_samples, K = am->num_iter, s = S / K;
  const int PL = prob->dof == 6 ? 7 : 4;
  MfmaShape sh;
  sh.s16 = ((s + 15) / 16) * 16;
  // register mode: W waves x NPT point tiles of 16 cover N with the least padding (ties: prefer 4 waves)
  const int ptiles = (d.N + 15) / 16;
  const int npt_opts[6] = {1, 2, 4, 8, 12, 16};
  int waves = 4, npt = 0, best = 1 << 30;
  for (int w : {4, 8, 2, 1}) {
    for (int o : npt_opts) {
      if (w * o >= ptiles && w * o < best) { best = w * o; waves = w; npt = o; }
      if (w * o >= ptiles) break;
    }
  }
  // 4-DoF with 12 or 16 resident tiles does not fit 256 VGPRs next to the von Mises sampler -- those instantiations spill
  // 120-308 B per lane and reload B-operand tuples INSIDE the sweep loop, the pattern of profiles/r05_bwd_scratch.txt --: stream the
  // points through LDS instead (12 tiles were taken until round 5; EPROPNP_TUNE=fwd_mfma=4,12 still selects them)
  if (prob->dof == 4 && npt >= 12) npt = 0;
  // few objects (less than two waves per SIMD otherwise): spread an object over 8 waves (B = 32: 86 vs 93 us)
  if (d.B < 512 && waves == 4 && npt == 8) { waves = 8; npt = 4; }
  { int ov[2]; if (tune_ints("fwd_mfma", ov, 2) && (ov[0] == 1 || ov[0] == 2 || ov[0] == 4 || ov[0] == 8) && (ov[1] == 0 || ov[1] == 1 || ov[1] == 2 || ov[1] == 4 || ov[1] == 8 || ov[1] == 12 || ov[1] == 16) && (ov[1] == 0 || ov[0] * ov[1] >= ptiles)) { waves = ov[0]; npt = ov[1]; } }
  if (npt == 0) {       // points stream through LDS in chunks; waves split the pose tiles
    sh.chunk = ((d.N + 15) / 16) * 16;
    if (sh.chunk > kChunk) sh.chunk = kChunk;
    if (best == (1 << 30)) waves = 4;
    const int tiles = sh.s16 / 16;
    while (waves > 1 && waves > tiles) waves /= 2;
  } else {
    sh.chunk = 0;
  }
  // Few objects (register mode): G workgroups per object, each sweeping every G-th group of point tiles (kernel comment).
  // (Decided on the tiles of a PART: 32 objects x 4096 points do not fit one workgroup's registers -- 0.415 ms streaming the
  // points through LDS on 32 CUs -- but an eighth of them does.)
  int G = 1;
  {
    const int g = forward_split_parts(d.B, ptiles);
    const size_t need = sizeof(float) * (size_t)d.B * K * g * sh.s16;
    int ovf[1];
    const bool forced = env_ints("EPROPNP_FWD_SPLIT", ovf, 1);           // experiments: scratch from hipMallocAsync
    if (g > 1 && (forced || (am->split_scratch != nullptr && am->split_scratch_bytes >= need))) {
      int per_wave = (ptiles + 4 * g - 1) / (4 * g), o = 1;      // tiles per wave, rounded up to an instantiated NPT
      while (o < per_wave) o *= 2;
      if (o <= 8) { G = g; waves = 4; npt = o; sh.chunk = 0; }
    }
  }
  // Many points per object in register mode (> 48 tiles: 16 resident tiles per wave or 8-wave workgroups, i.e. 256 VGPRs and one
  // or two workgroups per CU, whose serial sampler phases nothing hides): 4 waves x 8 resident tiles instead, the object's tiles
  // going through the registers in chunks of 32 per iteration -- three workgroups per CU again.  The points are re-read per
  // chunk and iteration (from L2 / HBM: 28 B per point against ~100 ns of arithmetic per 16 of them and pose tile).
  sh.chunks = 1;
  { const char* e = getenv("EPROPNP_FWD_PROJ"); if (e && e[0] == 'f') sh.chunks = 0; }      // (the chunked instantiation is the split projection)
  if (sh.chunks == 1 && G == 1 && npt > 8 && ptiles > 48 && !tune_flag("fwd_no_chunks") && !tune_flag("fwd_mfma")) {
    waves = 4; npt = 8;
    sh.chunks = (ptiles + 31) / 32;
  }
  if (sh.chunks < 1) sh.chunks = 1;
  AmisParams k;
  k.S = S; k.K = K; k.WP = 1; k.eps = am->eps; k.mle_iter = am->acg_mle_iter; k.dispersion = am->acg_dispersion;
  k.seed = am->seed; k.offset = am->offset; k.offset_dev = (const unsigned long long*)am->offset_dev; k.ablate = 0;
  k.advance = (am->advance && am->advance_ticket && am->advance_count > 0) ? (unsigned long long*)am->advance : nullptr;
  k.advance_ticket = (int*)am->advance_ticket; k.advance_count = am->advance_count;
  k.split_timeout = split_timeout_cycles();
  k.dn_offset = dn ? dn->offset : nullptr; k.dn_samples = dn ? dn->samples : nullptr; k.dn_pose_opt = dn ? dn->pose_opt : nullptr;
  { int ab[1]; if (tune_ints("ablate", ab, 1)) k.ablate = ab[0]; }
  sh.ahead = 1;
  auto lds_bytes = [&](bool spilled) {
    return sizeof(float) * (12 * (size_t)sh.s16 + 8 * (size_t)sh.chunk + (spilled ? 0 : (((size_t)(PL + 3) * S + 3) & ~(size_t)3)) +
                            (size_t)(npt ? (G > 1 ? waves + G : waves) : 1) * sh.s16 + (G > 1 ? 4 : 0) + (size_t)K * kPropStride + 256 +
                            kRefitRedFloats +
                            ((s <= 64 * waves || !sh.ahead) ? 0 : 8 * (size_t)s));      // (s <= lanes: the noise shares the pose table)
  };
  size_t smem = lds_bytes(false);
  float* spill = nullptr;
  if (smem > 160 * 1024) {
    // the sampler state does not fit LDS: stream the points (NPT = 0, 8 waves) and keep the per-sample arrays in a global
    // scratch buffer, allocated and released in stream order.  If one ITERATION's pose table (48 B per sample) and noise
    // buffer do not fit either, the noise is drawn inline and, if need be, the iteration's samples go through the table in
    // tiles (draw -> sweep -> costs to the scratch, per tile): no limit on mc_samples / num_iter (the reference has none,
    // epropnp.py:55-59)
    waves = 8; npt = 0; G = 1; sh.chunks = 1;
    sh.chunk = ((d.N + 15) / 16) * 16;
    if (sh.chunk > kChunk) sh.chunk = kChunk;
    const int tiles = sh.s16 / 16;
    while (waves > 1 && waves > tiles) waves /= 2;
    smem = lds_bytes(true);
    if (smem > 160 * 1024) {
      sh.ahead = 0;
      const size_t fixed = sizeof(float) * (8 * (size_t)sh.chunk + (size_t)K * kPropStride + 256 + kRefitRedFloats);
      const size_t rows = (160 * 1024 - fixed) / (sizeof(float) * 13);         // 12 pose-table floats + 1 cost per sample
      if ((size_t)sh.s16 > rows) sh.s16 = (int)(rows / 16) * 16;
      if (sh.s16 < 16) return fail(EPROPNP_EINVAL, "amis_forward: no LDS left for a pose tile (%d points per chunk)", sh.chunk);
      smem = lds_bytes(true);
    }
    if (hipMallocAsync((void**)&spill, sizeof(float) * (size_t)(PL + 3) * S * d.B, st) != hipSuccess) {
      (void)hipGetLastError();
      return fail(EPROPNP_ELAUNCH, "amis_forward: mc_samples %d needs a %zu B scratch buffer and hipMallocAsync failed "
                  "(stream capture?)", S, sizeof(float) * (size_t)(PL + 3) * S * d.B);
    }
  }
  // projection flavour (kernel comment): the bf16x3 split wherever the points are register-resident (C2 -9.8 %, 4096 x 1024 and the
  // C5 shard -16 %, profiles/r04_fwd_bf16_projection.txt); EPROPNP_FWD_PROJ=f32 keeps the fp32 MFMA
  bool bf16 = npt >= 1;
  if (const char* e = getenv("EPROPNP_FWD_PROJ")) bf16 = (e[0] == 'f') ? false : bf16;
  const dim3 grid(padded_object_grid(d.B)), block(64 * waves);
  if (spill != nullptr) {
    dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
      auto kern = amis_forward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, 0, true>;
      allow_dynamic_lds((const void*)kern, smem);
      PNP_LAUNCH(kern, grid, block, smem, st, d, k, sh, pose_opt, pose_cov, noise, pose_samples, logweights, proposals, spill, 1,
                 (float*)nullptr);
      return 0;
    });
    const int rc = check_launch("amis_forward_mfma_kernel (sampler state in global scratch)");
    (void)hipFreeAsync(spill, st);
    return rc;
  }
  float* xch = nullptr;
  char* owned = nullptr;
  dim3 grid_split = grid;
  if (G > 1) {
    // exchange buffer [B][K][G][s16], filled with the "not yet written" pattern on the stream: the caller's scratch (no
    // allocation: in a hipGraph an alloc / free node pair costs more than the split saves), or for EPROPNP_FWD_SPLIT
    // experiments a stream-ordered allocation
    const size_t xbytes = sizeof(float) * (size_t)d.B * K * G * sh.s16;
    char* mem = (am->split_scratch != nullptr && am->split_scratch_bytes >= xbytes) ? (char*)am->split_scratch : nullptr;
    if (mem == nullptr) {
      if (hipMallocAsync((void**)&mem, xbytes, st) != hipSuccess) {
        (void)hipGetLastError();
        return fail(EPROPNP_ELAUNCH, "amis_forward: scratch for the %d-way split (%zu B) could not be allocated", G, xbytes);
      }
      owned = mem;
    }
    if (!exchange_prefilled(mem, xbytes) &&
        launch_fill_u32(mem, 0xffffffffu, xbytes / 4, st) != EPROPNP_OK) {      // (a kernel, not a memset node: pnp_host.h)
      (void)hipGetLastError();
      return fail(EPROPNP_ELAUNCH, "amis_forward: could not fill the split scratch");
    }
    xch = (float*)mem;
    grid_split = dim3(padded_object_grid(d.B) * G);
  }
  if (G > 1) {
    dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
      auto go = [&](auto NPT) -> int {
        auto run = [&](auto kern) -> int {
          allow_dynamic_lds((const void*)kern, smem);
          PNP_LAUNCH(kern, grid_split, block, smem, st, d, k, sh, pose_opt, pose_cov, noise, pose_samples, logweights, proposals,
                     (float*)nullptr, G, xch);
          return 0;
        };
        if (bf16) return run(amis_forward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, decltype(NPT)::value, false, true, true>);
        return run(amis_forward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, decltype(NPT)::value, false, true>);
      };
      switch (npt) {
        case 1: return go(ic<1>{});
        case 2: return go(ic<2>{});
        case 4: return go(ic<4>{});
        default: return go(ic<8>{});
      }
    });
  } else {
    dispatch_dof_bounds(prob->dof, has_bounds(prob), [&](auto DOF, auto BND) -> int {
      return dispatch_npt(npt, [&](auto NPT) -> int {
        auto run = [&](auto kern) -> int {
          allow_dynamic_lds((const void*)kern, smem);
          PNP_LAUNCH(kern, grid, block, smem, st, d, k, sh, pose_opt, pose_cov, noise, pose_samples, logweights, proposals,
                     (float*)nullptr, 1, (float*)nullptr);
          return 0;
        };
        if constexpr (decltype(NPT)::value == 8) {
          if (sh.chunks > 1 || (bf16 && tune_flag("fwd_chunked")))      // (the knob: the chunked instantiation with one chunk)
            return run(amis_forward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, 8, false, false, true, true>);
        }
        if constexpr (decltype(NPT)::value >= 1) {
          if (bf16) return run(amis_forward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, decltype(NPT)::value, false, false, true>);
        }
        return run(amis_forward_mfma_kernel<decltype(DOF)::value, decltype(BND)::value, decltype(NPT)::value>);
      });
    });
  }
  if (owned != nullptr) (void)hipFreeAsync(owned, st);
  return check_launch("amis_forward_mfma_kernel");
}

}  // namespace pnp
