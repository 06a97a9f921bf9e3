// c_api.hip -- extern "C" surface of libepropnp_hip.so (declared in include/epropnp_hip.h).
#include <string.h>

#include <mutex>
#include <vector>

#include "pnp_host.h"

namespace pnp {
int tuning_phase_cycles(unsigned long long* out, int reset);          // amis_forward_mfma.hip / rslm_kernel.hip (tuning.h)
int tuning_rslm_phase_cycles(unsigned long long* out, int reset);
int tuning_bwd_phase_cycles(unsigned long long* out, int reset);
char* last_error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

// ---- per-stage HIP-event timing ------------------------------------------------------------------------------------
namespace {
struct StageRec { const char* stage; hipEvent_t e0, e1; bool closed; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<StageRec> g_prof;          // records in use
std::vector<hipEvent_t> g_pool;        // events created ahead of time: no hipEventCreate inside a timed region
constexpr size_t kMaxStageRecs = 1 << 15;
void fill_pool(size_t want) {
  while (g_pool.size() < want) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) break;
    g_pool.push_back(e);
  }
}
void drop_records() {
  for (auto& r : g_prof) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
  g_prof.clear();
}
}  // namespace
bool profile_begin(const char* stage, hipStream_t st) {
  if (!g_prof_on) return false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof.size() >= kMaxStageRecs) return false;
  if (g_pool.size() < 2) fill_pool(64);
  if (g_pool.size() < 2) return false;
  StageRec r{stage, g_pool[g_pool.size() - 1], g_pool[g_pool.size() - 2], false};
  g_pool.resize(g_pool.size() - 2);
  (void)hipEventRecord(r.e0, st);
  g_prof.push_back(r);
  return true;
}
void profile_end(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (size_t i = g_prof.size(); i-- > 0;)
    if (!g_prof[i].closed) { (void)hipEventRecord(g_prof[i].e1, st); g_prof[i].closed = true; break; }
}
}  // namespace pnp

namespace pnp {
// The default status word: when a caller passes no `epropnp_problem.status`, kernels report numerical events (a singular
// damped system, a non-finite pose: what torch.linalg.solve / torch.inverse raise on in the reference,
// levenberg_marquardt.py:15-19,178-181) into a per-device int32[2] in HOST memory mapped into the device.  It costs nothing
// on the normal path (no event -> no memory operation) and lets the host side notice a failure with a plain load -- no
// synchronisation, no copy: the Python layer polls it on entry to every call and raises what the reference would have
// raised, one call late at most (`epropnp_async_status`).  EPROPNP_ASYNC_STATUS=0 switches it off.
static int32_t* g_status_words[64] = {nullptr};
static int g_status_mode = -1;

int32_t* default_status_word() {
  if (g_status_mode < 0) {
    const char* e = getenv("EPROPNP_ASYNC_STATUS");
    g_status_mode = (e != nullptr && e[0] == '0') ? 0 : 1;
  }
  if (g_status_mode == 0) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  int32_t* w = __atomic_load_n(&g_status_words[dev], __ATOMIC_ACQUIRE);
  if (w != nullptr) return w;
  // may be the first call of a process that is already capturing a graph: a host allocation is not a stream operation,
  // relax the thread's capture mode around it
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  void* mem = nullptr;
  const hipError_t rc = hipHostMalloc(&mem, 2 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent);
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (rc != hipSuccess) { (void)hipGetLastError(); g_status_mode = 0; return nullptr; }
  w = (int32_t*)mem;
  w[0] = 0;
  w[1] = INT32_MAX;
  int32_t* expected = nullptr;
  if (!__atomic_compare_exchange_n(&g_status_words[dev], &expected, w, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
    (void)hipHostFree(w);
    w = expected;
  }
  return w;
}
}  // namespace pnp

extern "C" {

int32_t* epropnp_async_status_word(void) { return pnp::default_status_word(); }

int epropnp_async_status(int32_t* flags_and_first, int clear) {
  int32_t* w = pnp::default_status_word();
  const int32_t f = w ? __atomic_load_n(&w[0], __ATOMIC_RELAXED) : 0;
  const int32_t o = w ? __atomic_load_n(&w[1], __ATOMIC_RELAXED) : INT32_MAX;
  if (flags_and_first) { flags_and_first[0] = f; flags_and_first[1] = o; }
  if (w && clear) { __atomic_store_n(&w[0], 0, __ATOMIC_RELAXED); __atomic_store_n(&w[1], INT32_MAX, __ATOMIC_RELAXED); }
  return f;
}

int epropnp_abi_version(void) { return EPROPNP_ABI_VERSION; }

const char* epropnp_last_error(void) { return pnp::last_error_buffer(); }

uint64_t epropnp_amis_forward_split_bytes(const epropnp_problem* prob, int32_t mc_samples, int32_t num_iter) {
  return pnp::amis_forward_split_bytes(prob, mc_samples, num_iter);
}

int epropnp_noise_stride(int dof) { return dof == 6 ? 8 : (dof == 4 ? 4 + 3 * 16 : -1); }

int epropnp_monte_carlo_forward(const epropnp_problem* prob, const epropnp_mc_params* par, const float* pose_init,
                                const float* noise, float* x3d_centered, float* offset, float* pose_init_n,
                                float* start_pose, float* start_cost, float* pose_opt_n, float* pose_cov, float* cost,
                                float* pose_samples_n, float* logweights, float* cost_init, float* pose_opt,
                                float* pose_samples, void* stream) {
  return pnp::launch_monte_carlo_forward(prob, par, pose_init, noise, x3d_centered, offset, pose_init_n, start_pose,
                                         start_cost, pose_opt_n, pose_cov, cost, pose_samples_n, logweights, cost_init,
                                         pose_opt, pose_samples, (hipStream_t)stream);
}

int epropnp_evaluate_cost(const epropnp_problem* prob, const float* poses, int32_t num_poses, float* cost, void* stream) {
  pnp::StageScope prof_("evaluate_cost", (hipStream_t)stream);
  return pnp::launch_evaluate_cost(prob, poses, num_poses, cost, (hipStream_t)stream);
}

int epropnp_cost_pose_cam_grad(const epropnp_problem* prob, const float* poses, const float* weights, int32_t num_poses,
                               int32_t m_pose, float* grad_h_outer, float* grad_cam, void* stream) {
  pnp::StageScope prof_("cost_pose_cam_grad", (hipStream_t)stream);
  return pnp::launch_cost_pose_cam_grad(prob, poses, weights, num_poses, m_pose, grad_h_outer, grad_cam, (hipStream_t)stream);
}

int epropnp_normal_equations(const epropnp_problem* prob, const float* pose, int32_t clip_jac, float* jtj, float* jtr,
                             float* cost, void* stream) {
  pnp::StageScope prof_("normal_equations", (hipStream_t)stream);
  return pnp::launch_normal_equations(prob, pose, clip_jac, jtj, jtr, cost, (hipStream_t)stream);
}

int epropnp_lm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, const float* pose_init, float* pose_opt,
                     float* pose_cov, float* cost, int32_t* accept_mask, void* split_scratch, uint64_t split_scratch_bytes,
                     void* stream) {
  pnp::StageScope prof_("lm_solve", (hipStream_t)stream);
  return pnp::launch_lm_solve(prob, lm, pose_init, pose_opt, pose_cov, cost, accept_mask, split_scratch, split_scratch_bytes,
                              (hipStream_t)stream);
}

uint64_t epropnp_lm_solve_split_bytes(const epropnp_problem* prob, const epropnp_lm_params* lm) {
  return pnp::lm_split_bytes(prob, lm);
}

int epropnp_amis_forward(const epropnp_problem* prob, const epropnp_amis_params* amis, const float* pose_opt,
                         const float* pose_cov, const float* noise, float* pose_samples, float* logweights,
                         float* proposals, void* stream) {
  pnp::StageScope prof_("amis_forward", (hipStream_t)stream);
  return pnp::launch_amis_forward(prob, amis, pose_opt, pose_cov, noise, pose_samples, logweights, proposals,
                                  (hipStream_t)stream);
}

int epropnp_amis_backward(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                          int32_t mc_samples, const float* pose_init, const float* grad_cost_init, float* grad_x3d,
                          float* grad_x2d, float* grad_w2d, float* grad_delta, void* stream) {
  pnp::StageScope prof_("amis_backward", (hipStream_t)stream);
  return pnp::launch_amis_backward(prob, pose_samples, grad_logweights, mc_samples, pose_init, grad_cost_init, grad_x3d,
                                   grad_x2d, grad_w2d, grad_delta, (hipStream_t)stream);
}

int epropnp_amis_backward_split(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                                int32_t mc_samples, const float* pose_init, const float* grad_cost_init, int32_t num_split,
                                float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta_parts, void* stream) {
  pnp::StageScope prof_("amis_backward", (hipStream_t)stream);
  return pnp::launch_amis_backward_split(prob, pose_samples, grad_logweights, mc_samples, pose_init, grad_cost_init,
                                         num_split, grad_x3d, grad_x2d, grad_w2d, grad_delta_parts, (hipStream_t)stream);
}


int epropnp_adaptive_delta(const float* x2d, const float* w2d, int32_t num_obj, int32_t num_pts, float relative_delta,
                           float* delta, float* stats, void* stream) {
  pnp::StageScope prof_("adaptive_delta", (hipStream_t)stream);
  return pnp::launch_adaptive_delta(x2d, w2d, num_obj, num_pts, relative_delta, delta, stats, (hipStream_t)stream);
}

int epropnp_mc_loss_forward(const float* logweights, const float* cost_target, int32_t mc_samples, int32_t num_obj,
                            float* loss, float* lse, void* stream) {
  pnp::StageScope prof_("mc_loss_forward", (hipStream_t)stream);
  return pnp::launch_mc_loss_forward(logweights, cost_target, mc_samples, num_obj, loss, lse, (hipStream_t)stream);
}

int epropnp_mc_loss_backward(const float* logweights, const float* lse, const float* loss, const float* grad_loss,
                             int32_t mc_samples, int32_t num_obj, float* grad_logweights, float* grad_cost_target,
                             void* stream) {
  pnp::StageScope prof_("mc_loss_backward", (hipStream_t)stream);
  return pnp::launch_mc_loss_backward(logweights, lse, loss, grad_loss, mc_samples, num_obj, grad_logweights,
                                      grad_cost_target, (hipStream_t)stream);
}

int epropnp_mc_loss_reduce(const float* loss, const float* weight, int32_t num_obj, float scale, float momentum,
                           const float* norm_factor_in, int32_t norm_factor_in_count, int64_t norm_factor_in_stride,
                           float* norm_factor, float* out, void* stream) {
  pnp::StageScope prof_("mc_loss_reduce", (hipStream_t)stream);
  return pnp::launch_mc_loss_reduce(loss, weight, num_obj, scale, momentum, norm_factor_in, norm_factor_in_count,
                                    (long long)norm_factor_in_stride, norm_factor, out, (hipStream_t)stream);
}

int epropnp_exchange_pack(const float* rows, uint64_t row_floats, const float* scalars, int32_t n_scalars,
                          const float* sum_src, uint64_t sum_floats, float sum_scale, const float* sum_row_weight,
                          int32_t sum_row_len, float* send, void* stream) {
  return pnp::launch_exchange_pack(rows, (size_t)row_floats, scalars, n_scalars, sum_src, (size_t)sum_floats, sum_scale,
                                   sum_row_weight, sum_row_len, send, (hipStream_t)stream);
}

int epropnp_mc_loss_reduce_backward(const float* logweights, const float* lse, const float* weight, const float* coef,
                                    const float* grad_out, int32_t mc_samples, int32_t num_obj, float* grad_logweights,
                                    float* grad_cost_target, void* stream) {
  pnp::StageScope prof_("mc_loss_backward", (hipStream_t)stream);
  return pnp::launch_mc_loss_reduce_backward(logweights, lse, weight, coef, grad_out, mc_samples, num_obj, grad_logweights,
                                             grad_cost_target, (hipStream_t)stream);
}

int epropnp_rslm_draw(const float* w2d, int32_t num_obj, int32_t num_pts, int32_t num_proposals, int32_t n_pts,
                      uint64_t seed, uint64_t offset, int64_t* inds, void* stream) {
  return pnp::launch_rslm_draw(w2d, num_obj, num_pts, num_proposals, n_pts, seed, offset, (long long*)inds,
                               (hipStream_t)stream);
}

int epropnp_gn_step_forward(const epropnp_problem* prob, float eps, const float* pose, float* step, void* stream) {
  pnp::StageScope prof_("gn_step_forward", (hipStream_t)stream);
  return pnp::launch_gn_step_forward(prob, eps, pose, step, nullptr, (hipStream_t)stream);
}

int epropnp_gn_step_backward(const epropnp_problem* prob, float eps, const float* pose, const float* grad_step,
                             float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta, void* stream) {
  pnp::StageScope prof_("gn_step_backward", (hipStream_t)stream);
  return pnp::launch_gn_step_backward(prob, eps, pose, grad_step, nullptr, grad_x3d, grad_x2d, grad_w2d, grad_delta,
                                      (hipStream_t)stream);
}

int epropnp_shift_poses_backward(const float* pose, const float* offset, const float* grad_out, int32_t num_poses,
                                 int32_t num_obj, int32_t dof, float sign, float* grad_pose, void* stream) {
  return pnp::launch_shift_poses_backward(pose, offset, grad_out, num_poses, num_obj, dof, sign, grad_pose,
                                          (hipStream_t)stream);
}

int epropnp_pose_opt_plus_forward(const epropnp_problem* prob, float eps, const float* pose, float* pose_plus,
                                  void* stream) {
  pnp::StageScope prof_("gn_step_forward", (hipStream_t)stream);
  return pnp::launch_gn_step_forward(prob, eps, pose, nullptr, pose_plus, (hipStream_t)stream);
}

int epropnp_pose_opt_plus_backward(const epropnp_problem* prob, float eps, const float* pose, const float* grad_pose_plus,
                                   float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta, void* stream) {
  pnp::StageScope prof_("gn_step_backward", (hipStream_t)stream);
  return pnp::launch_gn_step_backward(prob, eps, pose, nullptr, grad_pose_plus, grad_x3d, grad_x2d, grad_w2d, grad_delta,
                                      (hipStream_t)stream);
}

int epropnp_rslm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, int32_t num_proposals,
                       int32_t num_points, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                       const int64_t* inds, const float* rot, float* pose, float* cost, void* scratch,
                       uint64_t scratch_bytes, void* stream) {
  pnp::StageScope prof_("rslm_solve", (hipStream_t)stream);
  return pnp::launch_rslm_solve(prob, lm, num_proposals, num_points, seed, offset, (const unsigned long long*)offset_dev,
                                (const long long*)inds, rot, pose, cost, scratch, scratch_bytes, (hipStream_t)stream);
}

uint64_t epropnp_rslm_solve_scratch_bytes(const epropnp_problem* prob, int32_t num_proposals) {
  return pnp::rslm_scratch_bytes(prob, num_proposals);
}

int epropnp_center_points(const float* x3d, int32_t num_obj, int32_t num_pts, float* offset, float* x3d_centered,
                          void* stream) {
  pnp::StageScope prof_("center_points", (hipStream_t)stream);
  return pnp::launch_center_points(x3d, num_obj, num_pts, offset, x3d_centered, (hipStream_t)stream);
}

int epropnp_shift_poses(const float* pose, const float* offset, int32_t num_poses, int32_t num_obj, int32_t dof,
                        float sign, float* out, void* stream) {
  pnp::StageScope prof_("shift_poses", (hipStream_t)stream);
  return pnp::launch_shift_poses(pose, offset, num_poses, num_obj, dof, sign, out, (hipStream_t)stream);
}

int epropnp_prepare_forward(const float* noc, const float* dim, const float* logits, const float* scale,
                            int32_t num_obj, int32_t num_pts, int32_t mode, float* x3d, float* w2d, float* stats,
                            void* stream) {
  return pnp::launch_prepare_forward(noc, dim, logits, scale, num_obj, num_pts, mode, x3d, w2d, stats, (hipStream_t)stream);
}

int epropnp_prepare_backward(const float* noc, const float* dim, const float* logits, const float* scale,
                             const float* stats, const float* grad_x3d, const float* grad_w2d, int32_t num_obj,
                             int32_t num_pts, int32_t mode, float* grad_noc, float* grad_dim, float* grad_logits,
                             float* grad_scale, void* stream) {
  return pnp::launch_prepare_backward(noc, dim, logits, scale, stats, grad_x3d, grad_w2d, num_obj, num_pts, mode, grad_noc,
                                      grad_dim, grad_logits, grad_scale, (hipStream_t)stream);
}

int epropnp_prepare_dense_forward(const float* noc_map, const float* dim, const float* logit_map, const float* scale,
                                  const float* box, const int64_t* inds, int32_t num_obj, int32_t num_pts, int32_t height,
                                  int32_t width, int32_t mode, float* x3d, float* x2d, float* w2d, float* stats,
                                  void* stream) {
  return pnp::launch_prepare_dense_forward(noc_map, dim, logit_map, scale, box, (const long long*)inds, num_obj, num_pts,
                                           height, width, mode, x3d, x2d, w2d, stats, (hipStream_t)stream);
}

int epropnp_prepare_dense_backward(const float* noc_map, const float* dim, const float* logit_map, const float* scale,
                                   const int64_t* inds, const float* stats, const float* grad_x3d, const float* grad_w2d,
                                   int32_t num_obj, int32_t num_pts, int32_t height, int32_t width, int32_t mode,
                                   float* grad_noc_map, float* grad_dim, float* grad_logit_map, float* grad_scale,
                                   void* stream) {
  return pnp::launch_prepare_dense_backward(noc_map, dim, logit_map, scale, (const long long*)inds, stats, grad_x3d, grad_w2d,
                                            num_obj, num_pts, height, width, mode, grad_noc_map, grad_dim, grad_logit_map,
                                            grad_scale, (hipStream_t)stream);
}

int epropnp_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(pnp::g_prof_mu);
  if (on) pnp::fill_pool(8192);        // 4096 stage launches before anything is created inside a timed region
  pnp::g_prof_on = on != 0;
  return EPROPNP_OK;
}

int epropnp_profile_reset(void) {
  std::lock_guard<std::mutex> lk(pnp::g_prof_mu);
  pnp::drop_records();
  return EPROPNP_OK;
}

int epropnp_profile_read(const char* stage, float* mean_ms, int32_t* count) {
  if (!stage || !mean_ms || !count) return pnp::fail(EPROPNP_EINVAL, "profile_read: NULL argument");
  *mean_ms = 0.f;
  *count = 0;
  std::lock_guard<std::mutex> lk(pnp::g_prof_mu);
  double sum = 0.0;
  for (auto& r : pnp::g_prof) {
    if (!r.closed || strcmp(r.stage, stage) != 0) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) return pnp::fail(EPROPNP_ELAUNCH, "profile_read: event synchronise failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    sum += ms;
    ++*count;
  }
  if (*count > 0) *mean_ms = (float)(sum / *count);
  return EPROPNP_OK;
}

// Not part of the ABI (tools/tune.py): per-phase shader-clock totals of amis_forward_mfma_kernel / rslm_solve_kernel in a tuning
// build (build.py -D PNP_TUNING); -1 in the product build, whose kernels carry no counters.
int epropnp_tuning_phase_cycles(unsigned long long* out, int reset) { return pnp::tuning_phase_cycles(out, reset); }
int epropnp_tuning_rslm_cycles(unsigned long long* out, int reset) { return pnp::tuning_rslm_phase_cycles(out, reset); }
int epropnp_tuning_bwd_cycles(unsigned long long* out, int reset) { return pnp::tuning_bwd_phase_cycles(out, reset); }

}  // extern "C"
