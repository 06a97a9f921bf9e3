// lm_core.h -- the Levenberg-Marquardt / Gauss-Newton iteration shared by lm_solve_kernel (lm_kernel.hip) and the
// fused RSLM initialiser (rslm_kernel.hip).  The caller supplies `sweep(pose, clip_jac, acc)`, which fills
// acc = [upper-tri J^T J | J^T r | cost] reduced over the object's points (every lane of the owning group gets it).
// Reference: epropnp/levenberg_marquardt.py:132-241 (solve, _lm_iter), :255-265 (pose_add).
#pragma once
#include "../../include/epropnp_hip.h"
#include "pnp_sweep.h"

namespace pnp {

struct LmParams {
  int num_iter, fast_mode;
  float min_diag, max_diag, min_rel_decrease, radius0, radius_max, eps;
  unsigned split_timeout;   // split over workgroups: shader cycles a part waits for a sibling's partial sums (wave_ops.h)
};

// Where the LM kernel takes an object's starting pose from when the random-sample initialiser ran split over `parts` workgroups per
// object (rslm_kernel.hip): the parts' best proposals cand[part][b][1 + pose_len] (cost first) and, optionally, a rival pose with
// its cost (force_init_solve=True with a given pose_init: the cheaper of the two, levenberg_marquardt.py:124-130).  The winner
// over the parts is what rslm_reduce_kernel picks (ties: the lowest part = the lowest proposal index); folded into the solve it
// saves that launch in the one-call forward.  cand == nullptr: the plain pose_init array.
struct StartSelect {
  const float* cand;
  const float* rival_pose;
  const float* rival_cost;
  int parts;
};

// acc (upper-tri JtJ | Jtr | cost)  ->  dense symmetric matrix
template <int DOF>
PNP_FN void unpack_h(const float (&acc)[NormalEq<DOF>::NV], float (&H)[DOF][DOF]) {
  int idx = 0;
#pragma unroll
  for (int i = 0; i < DOF; ++i)
#pragma unroll
    for (int j = i; j < DOF; ++j) {
      H[i][j] = acc[idx];
      H[j][i] = acc[idx];
      ++idx;
    }
}

// On return `pose` is the solution, `cur` the normal equations / cost at the last accepted (LM) or last evaluated
// (GN fast mode) point, bit i of `accepted_bits` says whether LM step i was accepted.
template <int DOF, class Sweep>
PNP_FN void lm_iterate(const LmParams& lm, Sweep&& sweep, float (&pose)[PoseLen<DOF>::value],
                       float (&cur)[NormalEq<DOF>::NV], int& accepted_bits, int& status_bits) {
  constexpr int PL = PoseLen<DOF>::value;
  constexpr int NH = NormalEq<DOF>::NH, NV = NormalEq<DOF>::NV;
  accepted_bits = 0;
  if (lm.fast_mode) {
    // Gauss-Newton (levenberg_marquardt.py:136-152): J^T J + eps I, no clip_jac; pose_cov / cost come from the
    // last EVALUATED (pre-update) point.
    for (int it = 0; it < lm.num_iter; ++it) {
      sweep(pose, false, cur);
      float H[DOF][DOF], g[DOF];
      ScaledFactor<DOF> f;
      unpack_h<DOF>(cur, H);
#pragma unroll
      for (int i = 0; i < DOF; ++i) {
        H[i][i] += lm.eps;
        g[i] = cur[NH + i];
      }
      if (!scaled_cholesky<DOF>(H, f)) status_bits |= EPROPNP_ST_LM_NOT_SPD;   // the reference's LU raises on a zero pivot
      scaled_solve<DOF>(f, g);
      float step[DOF], nxt[PL];
#pragma unroll
      for (int i = 0; i < DOF; ++i) step[i] = -g[i];
      pose_add<DOF>(pose, step, nxt);
#pragma unroll
      for (int i = 0; i < PL; ++i) pose[i] = nxt[i];
    }
    if (lm.num_iter == 0) sweep(pose, false, cur);
  } else {
    // trust-region LM (Ceres-style), levenberg_marquardt.py:154-169 + _lm_iter
    sweep(pose, true, cur);
    float radius = lm.radius0, decrease = 2.0f;
    for (int it = 0; it < lm.num_iter; ++it) {
      float H[DOF][DOF], Hlm[DOF][DOF], g[DOF], st[DOF];
      ScaledFactor<DOF> f;
      unpack_h<DOF>(cur, H);
#pragma unroll
      for (int i = 0; i < DOF; ++i) {
#pragma unroll
        for (int j = 0; j < DOF; ++j) Hlm[i][j] = H[i][j];
        // diagonal += clamp(diagonal, min, max) / radius + eps   (:210-211)
        const float d = H[i][i];
        Hlm[i][i] = d + (fminf(fmaxf(d, lm.min_diag), lm.max_diag) / radius + lm.eps);
        g[i] = cur[NH + i];
        st[i] = g[i];
      }
      if (!scaled_cholesky<DOF>(Hlm, f)) status_bits |= EPROPNP_ST_LM_NOT_SPD;
      scaled_solve<DOF>(f, st);   // st = Hlm^-1 g ; step = -st
      float step[DOF], pose_new[PL];
#pragma unroll
      for (int i = 0; i < DOF; ++i) step[i] = -st[i];
      pose_add<DOF>(pose, step, pose_new);

      float nxt[NV];
      sweep(pose_new, true, nxt);

      // model_cost_change = -step^T (H step / 2 + g)     (:225)
      float mcc = 0.f;
#pragma unroll
      for (int i = 0; i < DOF; ++i) {
        float hs = 0.f;
#pragma unroll
        for (int j = 0; j < DOF; ++j) hs = fmaf(H[i][j], step[j], hs);
        mcc -= step[i] * (0.5f * hs + g[i]);
      }
      const float model_change = mcc;
      const float rel = (cur[NV - 1] - nxt[NV - 1]) / model_change;
      const bool ok = (rel >= lm.min_rel_decrease) && (model_change > 0.0f);
      if (ok) {   // wave-uniform
        accepted_bits |= (1 << (it & 31));
#pragma unroll
        for (int i = 0; i < PL; ++i) pose[i] = pose_new[i];
#pragma unroll
        for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
        const float t3 = 2.0f * rel - 1.0f;
        radius = radius / fmaxf(1.0f - t3 * t3 * t3, 1.0f / 3.0f);
      }
      radius = fminf(fmaxf(radius, lm.eps), lm.radius_max);   // clamp applies to every object (:235)
      if (ok) {
        decrease = 2.0f;
      } else {
        radius = radius / decrease;   // reject path is not re-clamped in the same iteration (:239)
        decrease *= 2.0f;
      }
    }
  }

}

}  // namespace pnp
