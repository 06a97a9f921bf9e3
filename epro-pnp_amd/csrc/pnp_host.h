// pnp_host.h -- host-side plumbing shared by the launchers: argument checks, error reporting, workgroup
// shape selection.  No torch types anywhere: the library is a plain C ABI over device pointers.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/epropnp_hip.h"
#include "pnp_sweep.h"

namespace pnp {

char* last_error_buffer();   // thread-local, defined in c_api.hip
int32_t* default_status_word();   // per-device host-mapped int32[2] the kernels report into when the caller gives none (c_api.hip)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(EPROPNP_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return EPROPNP_OK;
}

inline int check_problem(const epropnp_problem* p) {
  if (p == nullptr) return fail(EPROPNP_EINVAL, "problem is NULL");
  if (p->dof != 4 && p->dof != 6) return fail(EPROPNP_EINVAL, "dof must be 4 or 6, got %d", p->dof);
  if (p->num_obj < 0 || p->num_pts < 0) return fail(EPROPNP_EINVAL, "negative sizes");
  if (p->num_obj > 0 && p->num_pts > 0 &&
      (!p->x3d || !p->x2d || !p->w2d || !p->cam_mats || !p->delta))
    return fail(EPROPNP_EINVAL, "NULL device pointer in problem");
  if (!(p->z_min >= 0.f)) return fail(EPROPNP_EINVAL, "z_min must be >= 0 (a depth clamp), got %g", (double)p->z_min);
  return EPROPNP_OK;
}

inline Problem to_device_problem(const epropnp_problem* p) {
  Problem d;
  d.x3d = p->x3d; d.x2d = p->x2d; d.w2d = p->w2d; d.cam = p->cam_mats;
  d.lb = p->lb; d.ub = p->ub; d.delta = p->delta;
  d.z_min = p->z_min; d.B = p->num_obj; d.N = p->num_pts;
  d.huber_eps = (p->huber_eps > 0.f) ? p->huber_eps : 1e-10f;
  d.inv_huber_eps = 1.0f / d.huber_eps;
  d.status = p->status ? p->status : default_status_word();
  d.delta_stats = p->delta_stats; d.delta_relative = p->delta_relative;
  return d;
}

inline bool has_bounds(const epropnp_problem* p) { return p->lb != nullptr && p->ub != nullptr; }

constexpr int kMaxResidentPoints = 64 * 16 * 8;   // 16 waves x 64 lanes x 8 points per lane

// Pick waves-per-object and points-per-lane for kernels that keep the object's points in registers.
// Few objects -> more waves per object (fill the 1024 SIMDs); many objects -> fewer, fatter waves.
inline Shape choose_shape(int B, int N, int max_ppl = 8, int want_waves_total = 4096) {
  Shape s;
  int wmin = 1;
  while (64 * wmin * max_ppl < N && wmin < 16) wmin *= 2;
  // 16-wave (1024-thread) groups are limited to 128 VGPRs per lane and spill: only used when N demands it
  int wmax = 1;
  while (64 * wmax < N && wmax < 8) wmax *= 2;
  if (wmax < wmin) wmax = wmin;
  int w = wmin;
  while (w < wmax && (long)B * w < want_waves_total) w *= 2;
  s.waves = w;
  int ppl = 1;
  while (64 * w * ppl < N) ppl *= 2;
  s.ppl = ppl;
  return s;
}

// a (waves, points-per-lane) override from a tuning variable is taken only if it is one the kernels are instantiated for
// and covers the object: anything else is ignored (a typo must not change results)
inline bool valid_shape_override(int waves, int ppl, int N) {
  const bool w_ok = waves == 1 || waves == 2 || waves == 4 || waves == 8 || waves == 16;
  const bool p_ok = ppl == 1 || ppl == 2 || ppl == 4 || ppl == 8;
  return w_ok && p_ok && 64L * waves * ppl >= N;
}

inline bool parse_ints(const char* v, int* out, int n) {
  if (!v || !*v) return false;
  for (int i = 0; i < n; ++i) {
    char* end = nullptr;
    out[i] = (int)strtol(v, &end, 10);
    if (end == v) return false;
    v = (*end == ',') ? end + 1 : end;
  }
  return true;
}

// A user-facing knob EPROPNP_<NAME>="a[,b]" (INTEGRATION.md lists them all: *_PROJ, BWD_DROP, *_SPLIT, ASYNC_STATUS, NO_TORCH_EXT,
// DELTA_FOLD, LOSS_FUSED, SPLIT_TIMEOUT_CYCLES).
inline bool env_ints(const char* name, int* out, int n) { return parse_ints(getenv(name), out, n); }

// Everything else that used to be an environment variable of its own -- launch-shape overrides, implementation selectors,
// phase ablation: what tools/tune.py and the shape tests need, not what a user does -- is ONE string:
//     EPROPNP_TUNE="lm_shape=1,8;bwd_impl=valu;rslm_composite"
// `tune_value(key)` returns the text behind "key=" ("" for a bare key), or nullptr when the key is absent.
inline const char* tune_value(const char* key) {
  static thread_local char buf[64];
  const char* s = getenv("EPROPNP_TUNE");
  if (!s) return nullptr;
  const size_t kl = strlen(key);
  while (*s) {
    const char* end = strchr(s, ';');
    const size_t len = end ? (size_t)(end - s) : strlen(s);
    if (len >= kl && strncmp(s, key, kl) == 0 && (len == kl || s[kl] == '=')) {
      const size_t vl = (len == kl) ? 0 : len - kl - 1;
      if (vl >= sizeof(buf)) return nullptr;
      memcpy(buf, s + kl + (len == kl ? 0 : 1), vl);
      buf[vl] = 0;
      return buf;
    }
    if (!end) break;
    s = end + 1;
  }
  return nullptr;
}
inline bool tune_ints(const char* key, int* out, int n) { return parse_ints(tune_value(key), out, n); }
inline bool tune_flag(const char* key) { return tune_value(key) != nullptr; }

// Compute units of the current device (hipDeviceAttributeMultiprocessorCount, cached): what the workgroup-split variants
// size their grids against -- a partitioned (CPX) MI355X or another part reports its own count; the CPU emulation of the
// tests is a "device" with ONE compute unit (it runs one workgroup at a time), which switches the splits off by default.
inline int device_cu_count() {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

// How long (shader cycles) a part of a split kernel waits for a sibling's exchange words before it recomputes them itself
// (wave_ops.h: xwg_poll).  Default 2^18 (~0.1 ms: siblings normally arrive within ~2 us); EPROPNP_SPLIT_TIMEOUT_CYCLES
// overrides, 0 = never wait (every part recomputes whatever is not there yet: the tests' way of forcing that path).
inline unsigned split_timeout_cycles() {
  int ov[1];
  if (env_ints("EPROPNP_SPLIT_TIMEOUT_CYCLES", ov, 1) && ov[0] >= 0) return (unsigned)ov[0];
  return 1u << 18;
}

// kernels with more than 64 KiB of dynamic LDS have to be told so once
inline void allow_dynamic_lds(const void* kern, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Fraction of an object's total |weight| that the backward may leave out (see mass_drop_threshold, amis_common.h).
// Default 2^-24; EPROPNP_BWD_DROP=<float> overrides it, 0 = exact (every non-zero sample is evaluated).
inline float backward_drop_eps() {
  const char* v = getenv("EPROPNP_BWD_DROP");
  if (v && *v) {
    char* end = nullptr;
    const float f = strtof(v, &end);
    if (end != v && f >= 0.f && f < 1.f) return f;
  }
  return 5.9604644775390625e-08f;
}

// Optional per-stage timing (epropnp_profile_*, include/epropnp_hip.h): HIP events on the launch stream around each
// kernel stage, recorded inside the library so that stages launched from epropnp_monte_carlo_forward are visible too.
bool profile_begin(const char* stage, hipStream_t st);     // false (and no work) unless profiling is enabled
void profile_end(hipStream_t st);
struct StageScope {
  hipStream_t st;
  bool on;
  StageScope(const char* stage, hipStream_t s) : st(s), on(profile_begin(stage, s)) {}
  ~StageScope() { if (on) profile_end(st); }
};

// launchers (one per .hip translation unit)
int launch_monte_carlo_forward(const epropnp_problem* prob, const epropnp_mc_params* par, const float* pose_init,
                               const float* noise, float* x3d_centered, float* offset, float* pose_init_n,
                               float* start_pose, float* start_cost, float* pose_opt_n, float* pose_cov, float* cost,
                               float* pose_samples_n, float* logweights, float* cost_init, float* pose_opt,
                               float* pose_samples, hipStream_t st);
unsigned long long amis_forward_split_bytes(const epropnp_problem* prob, int mc_samples, int num_iter);
int launch_evaluate_cost(const epropnp_problem* prob, const float* poses, int num_poses, float* cost, hipStream_t st);
int launch_cost_pose_cam_grad(const epropnp_problem* prob, const float* poses, const float* weights, int num_poses,
                              int m_pose, float* out_m, float* out_gk, hipStream_t st);
int launch_normal_equations(const epropnp_problem* prob, const float* pose, int clip_jac, float* jtj, float* jtr,
                            float* cost, hipStream_t st);
int launch_amis_backward_mfma(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                              int mc_samples, const float* pose_init, const float* grad_cost_init, float* grad_x3d,
                              float* grad_x2d, float* grad_w2d, float* grad_delta, int nsplit, hipStream_t st);
int launch_rslm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, int P, int n_pts, unsigned long long seed,
                      unsigned long long offset, const unsigned long long* offset_dev, const long long* inds, const float* rot,
                      float* pose_out, float* cost_out, void* scratch, unsigned long long scratch_bytes,
                      hipStream_t st, const float* rival_pose = nullptr, const float* rival_cost = nullptr,
                      bool* rival_taken = nullptr, int* deferred_parts = nullptr);
unsigned long long rslm_scratch_bytes(const epropnp_problem* prob, int num_proposals);
int launch_shift_poses_pair(const float* pose_a, float* out_a, int Pa, const float* pose_b, float* out_b, int Pb,
                            const float* offset, int B, int dof, float sign, hipStream_t st);
int launch_center_points(const float* x3d, int B, int N, float* offset, float* out, hipStream_t st);
int launch_center_cost(const epropnp_problem* prob, const float* pose, float* offset, float* x3d_centered, float* pose_n,
                       float* cost, hipStream_t st);
int launch_center_points_shift(const float* x3d, int B, int N, float* offset, float* out, const float* pose, float* pose_out,
                               int dof, hipStream_t st);
int launch_shift_poses(const float* pose, const float* offset, int P, int B, int dof, float sign, float* out,
                       hipStream_t st);
int launch_prepare_forward(const float* noc, const float* dim, const float* logits, const float* scale, int B, int N,
                           int mode, float* x3d, float* w2d, float* stats, hipStream_t st);
int launch_prepare_backward(const float* noc, const float* dim, const float* logits, const float* scale, const float* stats,
                            const float* gx3d, const float* gw2d, int B, int N, int mode, float* gnoc, float* gdim,
                            float* glogits, float* gscale, hipStream_t st);
int launch_prepare_dense_forward(const float* noc_map, const float* dim, const float* logit_map, const float* scale,
                                 const float* box, const long long* inds, int B, int N, int H, int W, int mode, float* x3d,
                                 float* x2d, float* w2d, float* stats, hipStream_t st);
int launch_prepare_dense_backward(const float* noc_map, const float* dim, const float* logit_map, const float* scale,
                                  const long long* inds, const float* stats, const float* gx3d, const float* gw2d, int B,
                                  int N, int H, int W, int mode, float* gnoc_map, float* gdim, float* glogit_map,
                                  float* gscale, hipStream_t st);
int launch_shift_poses_backward(const float* pose, const float* offset, const float* gout, int P, int B, int dof, float sign,
                                float* gpose, hipStream_t st);
int launch_gn_step_forward(const epropnp_problem* prob, float eps, const float* pose, float* step, float* pose_plus,
                           hipStream_t st);
int launch_gn_step_backward(const epropnp_problem* prob, float eps, const float* pose, const float* grad_step,
                            const float* grad_pose_plus, float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta, hipStream_t st);
int launch_rslm_draw(const float* w2d, int B, int N, int P, int n_pts, unsigned long long seed, unsigned long long offset,
                     long long* inds, hipStream_t st);
int launch_adaptive_delta(const float* x2d, const float* w2d, int B, int N, float rel, float* delta, float* stats,
                          hipStream_t st);
int launch_mc_loss_forward(const float* logw, const float* ct, int S, int B, float* loss, float* lse, hipStream_t st);
// grad_w2d += grad_delta * d delta / d w2d for a threshold from AdaptiveHuberPnPCost (epropnp_problem.delta_stats); no-op without
int launch_delta_path(const epropnp_problem* prob, const float* gdelta, int nparts, float* gw2d, hipStream_t st);
// stream-ordered fill as a kernel (never hipMemsetAsync: eval_kernels.hip, fill_u32_kernel)
int launch_fill_u32(void* p, unsigned v, size_t words, hipStream_t st);
// The exchange scratch of the split LM solve and of the split AMIS forward start out as 0xffffffff words ("not yet written").  The
// one-call forward fills BOTH with one launch when the caller hands them over as one contiguous block (mc_forward.hip) and marks
// the range here; a launcher whose scratch lies inside the marked range skips its own fill.  Thread-local, valid for the duration
// of that one host call.
struct PrefilledRange { const char* lo; const char* hi; };
PrefilledRange& prefilled_exchange_range();      // mc_forward.hip
inline bool exchange_prefilled(const void* p, size_t bytes) {
  const PrefilledRange& r = prefilled_exchange_range();
  return r.lo != nullptr && (const char*)p >= r.lo && (const char*)p + bytes <= r.hi;
}
int launch_mc_loss_reduce(const float* loss, const float* weight, int B, float scale, float momentum, const float* nf_in,
                          int nf_count, long long nf_stride, float* nf, float* out, hipStream_t st);
int launch_exchange_pack(const float* rows, size_t row_floats, const float* scalars, int n_scal, const float* sum_src,
                         size_t sum_floats, float sum_scale, const float* row_w, int row_len, float* send, hipStream_t st);
int launch_mc_loss_reduce_backward(const float* logw, const float* lse, const float* weight, const float* coef,
                                   const float* gout, int S, int B, float* glogw, float* gct, hipStream_t st);
int launch_mc_loss_backward(const float* logw, const float* lse, const float* loss, const float* g, int S, int B,
                            float* glogw, float* gct, hipStream_t st);
struct StartSelect;      // lm_core.h
int launch_lm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, const float* pose_init, float* pose_opt,
                    float* pose_cov, float* cost, int32_t* accept_mask, void* split_scratch,
                    unsigned long long split_scratch_bytes, hipStream_t st, const StartSelect* select = nullptr);
unsigned long long lm_split_bytes(const epropnp_problem* prob, const epropnp_lm_params* lm);
// (one-call forward on a pnp_normalize'd problem: the AMIS launch also writes pose_opt and the samples in the caller's frame --
//  AmisParams.dn_* -- instead of a shift_poses_pair launch behind it)
struct DenormOut { const float* offset; float* samples; float* pose_opt; };
int launch_amis_forward(const epropnp_problem* prob, const epropnp_amis_params* amis, const float* pose_opt,
                        const float* pose_cov, const float* noise, float* pose_samples, float* logweights,
                        float* proposals, hipStream_t st, const DenormOut* dn = nullptr);
int launch_amis_forward_mfma(const epropnp_problem* prob, const epropnp_amis_params* amis, const float* pose_opt,
                             const float* pose_cov, const float* noise, float* pose_samples, float* logweights,
                             float* proposals, hipStream_t st, const DenormOut* dn = nullptr);
int launch_amis_backward_split(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                               int mc_samples, const float* pose_init, const float* grad_cost_init, int nsplit,
                               float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta_parts, hipStream_t st);
int launch_amis_backward(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                         int mc_samples, const float* pose_init, const float* grad_cost_init, float* grad_x3d,
                         float* grad_x2d, float* grad_w2d, float* grad_delta, hipStream_t st);

}  // namespace pnp
