// mc_forward.hip -- everything EProPnPBase.monte_carlo_forward does (epropnp/epropnp.py:87-196) behind ONE host call.
//
// The kernels are the ones the separate entry points launch (center_points / shift_poses, evaluate_cost, rslm_solve,
// lm_solve, amis_forward); what this entry removes is the host work between them.  At the launch-bound shapes
// (EPro-PnP-Det: 600 objects x 128 points; LineMOD training: 32 x 512) a step through the Python API spent 0.3-0.4 ms on
// the host -- allocator calls, ctypes marshalling, autograd-node bookkeeping around each of ~10 launches -- against
// ~0.2 ms of GPU work.  Stream-ordered and free of host synchronisation, so it is hipGraph-capturable like its parts.
#include "dispatch.h"
#include "lm_core.h"
#include "pnp_host.h"

namespace pnp {

PrefilledRange& prefilled_exchange_range() {
  static thread_local PrefilledRange r = {nullptr, nullptr};
  return r;
}
namespace {
struct PrefillScope {      // clears the mark when the host call returns, whichever way
  ~PrefillScope() { prefilled_exchange_range() = PrefilledRange{nullptr, nullptr}; }
};
}  // namespace

// force_init_solve=True with a given pose_init: per object the cheaper of {pose_init, RSLM pose}
// (levenberg_marquardt.py:124-130: `use_init = cost_init < cost_init_solve`), written over the RSLM pose.
__global__ __launch_bounds__(256) void select_start_kernel(const float* __restrict__ pose_init,
                                                           const float* __restrict__ cost_init,
                                                           float* __restrict__ start_pose,
                                                           const float* __restrict__ start_cost, int B, int PL) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= B * PL) return;
  const int b = i / PL;
  if (cost_init[b] < start_cost[b]) start_pose[i] = pose_init[i];
}

int launch_monte_carlo_forward(const epropnp_problem* prob, const epropnp_mc_params* par, const float* pose_init,
                               const float* noise, float* x3d_centered, float* offset, float* pose_init_n,
                               float* start_pose, float* start_cost, float* pose_opt_n, float* pose_cov, float* cost,
                               float* pose_samples_n, float* logweights, float* cost_init, float* pose_opt,
                               float* pose_samples, hipStream_t st) {
  if (int rc = check_problem(prob)) return rc;
  if (!par) return fail(EPROPNP_EINVAL, "monte_carlo_forward: params NULL");
  if (prob->num_obj == 0) return EPROPNP_OK;
  const int B = prob->num_obj, PL = prob->dof == 6 ? 7 : 4, S = par->amis.mc_samples;
  if (!pose_opt_n || !pose_cov || !pose_samples_n || !logweights)
    return fail(EPROPNP_EINVAL, "monte_carlo_forward: NULL output pointer");
  if (par->init_mode < 0 || par->init_mode > 2) return fail(EPROPNP_EINVAL, "monte_carlo_forward: init_mode %d", par->init_mode);
  if ((par->init_mode == 1) == (pose_init != nullptr))
    return fail(EPROPNP_EINVAL, "monte_carlo_forward: init_mode %d %s pose_init", par->init_mode,
                par->init_mode == 1 ? "takes no" : "needs a");
  if (pose_init && !cost_init) return fail(EPROPNP_EINVAL, "monte_carlo_forward: cost_init NULL with a pose_init");
  if (par->init_mode != 0 && (!start_pose || !start_cost))
    return fail(EPROPNP_EINVAL, "monte_carlo_forward: start_pose / start_cost scratch NULL");
  if (par->normalize && (!x3d_centered || !offset || !pose_opt || !pose_samples || (pose_init && !pose_init_n)))
    return fail(EPROPNP_EINVAL, "monte_carlo_forward: normalize needs x3d_centered, offset, pose_init_n, pose_opt, pose_samples");

  epropnp_problem q = *prob;
  const float* pinit = pose_init;
  int rc;
  // both split kernels' exchange scratch in one contiguous block (LM first): ONE fill launch for the two (32 crops x 4096
  // points: a launch less in a step of ~12)
  PrefillScope prefill_scope;
  if (par->lm_scratch != nullptr && par->amis.split_scratch != nullptr && par->lm_scratch_bytes > 0 && par->amis.split_scratch_bytes > 0 &&
      (const char*)par->lm_scratch + par->lm_scratch_bytes == (const char*)par->amis.split_scratch &&
      (par->lm_scratch_bytes % 4) == 0 && (par->amis.split_scratch_bytes % 4) == 0) {
    const size_t total = (size_t)par->lm_scratch_bytes + (size_t)par->amis.split_scratch_bytes;
    if ((rc = launch_fill_u32(par->lm_scratch, 0xffffffffu, total / 4, st))) return rc;
    prefilled_exchange_range() = PrefilledRange{(const char*)par->lm_scratch, (const char*)par->lm_scratch + total};
  }
  bool have_cost_init = false;
  if (par->normalize) {       // pnp_normalize (common.py:103-124)
    {   // centred points and, in the same launch, pose_init in the centred frame -- and its cost there (:121-124), where the
        // object's points fit the registers of one workgroup
      StageScope ps("center_points", st);
      if (pose_init && prob->num_pts <= kMaxResidentPoints) {
        rc = launch_center_cost(prob, pose_init, offset, x3d_centered, pose_init_n, cost_init, st);
        have_cost_init = true;
      } else {
        rc = pose_init ? launch_center_points_shift(prob->x3d, B, prob->num_pts, offset, x3d_centered, pose_init, pose_init_n, prob->dof, st)
                       : launch_center_points(prob->x3d, B, prob->num_pts, offset, x3d_centered, st);
      }
      if (rc) return rc;
    }
    q.x3d = x3d_centered;
    if (pose_init) pinit = pose_init_n;
  }
  if (pinit && !have_cost_init) {                // cost of pose_init (:121-124)
    StageScope ps("evaluate_cost", st);
    if ((rc = launch_evaluate_cost(&q, pinit, 1, cost_init, st))) return rc;
  }
  const float* start = pinit;
  StartSelect sel;            // RSLM split over workgroups: its reduce (and the cheaper-of-two selection) runs inside the LM launch
  sel.cand = nullptr; sel.rival_pose = nullptr; sel.rival_cost = nullptr; sel.parts = 0;
  bool selected = false;      // the cheaper-of-two selection was folded into the RSLM reduce launch
  if (par->init_mode != 0) {  // random-sample initialiser (levenberg_marquardt.py:115-130,283-353)
    StageScope ps("rslm_solve", st);
    if ((rc = launch_rslm_solve(&q, &par->rslm_lm, par->rslm_proposals, par->rslm_points, par->rslm_seed, par->rslm_offset,
                                (const unsigned long long*)par->rslm_offset_dev, (const long long*)par->rslm_inds,
                                par->rslm_rot, start_pose, start_cost, par->rslm_scratch, par->rslm_scratch_bytes, st,
                                par->init_mode == 2 ? pinit : nullptr, par->init_mode == 2 ? cost_init : nullptr, &selected,
                                &sel.parts)))
      return rc;
    if (sel.parts >= 1) {     // no reduce / select launch was made: the LM kernel picks the winner itself
      sel.cand = (const float*)par->rslm_scratch;
      if (par->init_mode == 2) { sel.rival_pose = pinit; sel.rival_cost = cost_init; }
    }
    if (par->init_mode == 2 && !selected) {
      PNP_LAUNCH(select_start_kernel, dim3((B * PL + 255) / 256), dim3(256), 0, st, pinit, cost_init, start_pose, start_cost,
                 B, PL);
      if ((rc = check_launch("select_start_kernel"))) return rc;
    }
    start = start_pose;
  }
  { StageScope ps("lm_solve", st); if ((rc = launch_lm_solve(&q, &par->lm, start, pose_opt_n, pose_cov, cost, nullptr, par->lm_scratch, par->lm_scratch_bytes, st, sel.cand ? &sel : nullptr))) return rc; }
  {   // normalize: pnp_denormalize (common.py:127-136) of pose_opt and of the samples rides in the AMIS launch
    const DenormOut dn = {offset, pose_samples, pose_opt};
    const bool fold = par->normalize && !tune_flag("no_denorm_fold");
    StageScope ps("amis_forward", st);
    if ((rc = launch_amis_forward(&q, &par->amis, pose_opt_n, pose_cov, noise, pose_samples_n, logweights, nullptr, st, fold ? &dn : nullptr)))
      return rc;
    if (fold) return EPROPNP_OK;
  }
  if (par->normalize) {       // (EPROPNP_TUNE=no_denorm_fold: the separate launch, same bits)
    StageScope ps("shift_poses", st);
    if ((rc = launch_shift_poses_pair(pose_opt_n, pose_opt, 1, pose_samples_n, pose_samples, S, offset, B, prob->dof, -1.0f, st)))
      return rc;
  }
  return EPROPNP_OK;
}

}  // namespace pnp
