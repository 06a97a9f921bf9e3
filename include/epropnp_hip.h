/*
 * epropnp_hip.h -- C ABI of libepropnp_hip.so, the MI355X (gfx950) implementation of the EPro-PnP hot path.
 *
 * The reference (tjiiv-cprg/EPro-PnP) has no FFI layer: its boundary is the Python class API of `epropnp/`.
 * Each entry point below replaces a span of that Python code (cited per function, paths relative to the
 * reference checkout); the Python classes in epro-pnp_amd/epropnp/ keep the reference's names/signatures and
 * call these through ctypes (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (HBM), owned by the caller; outputs are caller-allocated;
 *   - `stream` is a hipStream_t (passed as void*); the call enqueues work on it and returns without synchronising;
 *   - return value: 0 on success, negative EPROPNP_E* code on error; epropnp_last_error() gives the message
 *     (thread-local).  Nothing is ever computed on the host: without a HIP device the calls fail.
 *   - dof is 6 (pose = [x,y,z, w,i,j,k], unit quaternion) or 4 (pose = [x,y,z, yaw], rotation about Y).
 *   - B objects, N points per object; x3d (B,N,3), x2d (B,N,2), w2d (B,N,2).
 */
#ifndef EPROPNP_HIP_H
#define EPROPNP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPROPNP_OK 0
#define EPROPNP_EINVAL (-1)   /* bad argument (null pointer, unsupported dof, size limit)        */
#define EPROPNP_ELAUNCH (-2)  /* HIP launch/runtime error                                         */
#define EPROPNP_ENODEV (-3)   /* no HIP device / not a gfx950 code object                         */

#define EPROPNP_ABI_VERSION 6

/* Correspondences + camera + robust-cost parameters of one batch of objects.
 * Mirrors the state of PerspectiveCamera (epropnp/camera.py:35-62) and HuberPnPCost.delta
 * (epropnp/cost_fun.py:25-31,123-126) after the callers' set_param(). */
typedef struct epropnp_problem {
  const float* x3d;      /* (B,N,3) */
  const float* x2d;      /* (B,N,2) */
  const float* w2d;      /* (B,N,2) */
  const float* cam_mats; /* (B,3,3) row-major */
  const float* lb;       /* (B,2) lower bound [x,y] of the projection clamp, or NULL (camera.py:81-93)  */
  const float* ub;       /* (B,2) upper bound, or NULL; the clamp is applied only if both are non-NULL  */
  const float* delta;    /* (B,) Huber threshold per object                                             */
  float z_min;           /* camera.py:16,28                                                             */
  int32_t num_obj;       /* B */
  int32_t num_pts;       /* N */
  int32_t dof;           /* 6 or 4 */
  float huber_eps;       /* HuberPnPCost.eps (cost_fun.py:18-22,29): floor of |r| in the robust rescaling
                            sqrt(min(delta / max(|r|, eps), 1)); <= 0 selects the reference default 1e-10            */
  int32_t* status;       /* optional DEVICE int32[2] (or NULL): kernels OR EPROPNP_ST_* flags into status[0] and
                            atomicMin the first offending object index into status[1] (initialise to {0, INT32_MAX}).
                            Lets a caller reproduce, after a synchronisation of its choosing, the RuntimeError that
                            torch.linalg.solve / torch.inverse raise in the reference (levenberg_marquardt.py:15-19,
                            :178-181) instead of receiving NaN poses silently.  NULL selects the library's default
                            status word (epropnp_async_status below).                                                 */
  const float* delta_stats;  /* optional: the (B,4) `stats` of the epropnp_adaptive_delta call that produced `delta` from THIS
                            w2d (AdaptiveHuberPnPCost.set_param, cost_fun.py:123-126), else NULL.  Then d delta[b] / d w2d[b,n,c]
                            is the same number for every n, c -- stats[b][1] * delta_relative / (2 N) -- and the entry points
                            that return grad_w2d together with grad_delta (epropnp_amis_backward[_split], epropnp_gn_step_backward,
                            epropnp_pose_opt_plus_backward) add grad_delta[b] times it to every element of grad_w2d[b] themselves
                            (the backward kernel's epilogue, or one follow-up launch) instead of leaving three elementwise
                            launches and a (B,N,2) add to the caller's autograd; the caller must then NOT propagate grad_delta to
                            w2d again.  Forward entry points ignore it.                                                    */
  float delta_relative;      /* the relative_delta of that call (used only with delta_stats)                              */
} epropnp_problem;

/* status[0] flags */
#define EPROPNP_ST_LM_NOT_SPD 1        /* a damped normal-equation system had no Cholesky factor (singular / NaN input) */
#define EPROPNP_ST_NONFINITE_POSE 2    /* lm_solve / rslm_solve / gn_step produced a non-finite pose                    */
#define EPROPNP_ST_CHOL_FALLBACK 4     /* a proposal covariance was replaced by its default (cholesky_wrapper,
                                          epropnp.py:16-33: what the reference does silently as well)                  */
#define EPROPNP_ST_NONFINITE_WEIGHT 8  /* an AMIS log-weight is NaN / +inf                                              */
#define EPROPNP_ST_SPLIT_TIMEOUT 16    /* PERFORMANCE event, results unaffected: in a launch that splits an object over several
                                          workgroups (amis_forward at few objects, lm_solve beyond 2048 points) a part did
                                          not see a sibling's partial sums within EPROPNP_SPLIT_TIMEOUT_CYCLES (the
                                          siblings were not all resident: CU mask, partitioned GPU, a foreign kernel holding
                                          CUs) and recomputed them itself from the object's points, to the same bits     */

/* Trust-region parameters of LMSolver.__init__ (epropnp/levenberg_marquardt.py:31-53). */
typedef struct epropnp_lm_params {
  int32_t num_iter;
  int32_t fast_mode;                 /* 1: Gauss-Newton, no trust region, no clip_jac (:136-152) */
  float min_lm_diagonal;
  float max_lm_diagonal;
  float min_relative_decrease;
  float initial_trust_region_radius;
  float max_trust_region_radius;
  float eps;
} epropnp_lm_params;

/* AMIS parameters of EProPnPBase/EProPnP6DoF.__init__ (epropnp/epropnp.py:47-62,273-280). */
typedef struct epropnp_amis_params {
  int32_t mc_samples;     /* S, multiple of num_iter */
  int32_t num_iter;       /* K                        */
  float eps;              /* 1e-5                     */
  int32_t acg_mle_iter;   /* 3     (6-DoF only)       */
  float acg_dispersion;   /* 0.001 (6-DoF only)       */
  uint64_t seed;          /* Philox key when `noise` is NULL */
  uint64_t offset;        /* Philox counter offset (advance by 1 per call for fresh draws) */
  const uint64_t* offset_dev; /* optional DEVICE counter added to `offset` when the kernel runs: lets a captured hipGraph
                                 draw fresh samples on every replay (the caller increments it inside the graph) */
  void* split_scratch;        /* optional DEVICE scratch of `split_scratch_bytes` bytes (or NULL / 0).  With few objects
                                 (one workgroup per object would leave most CUs idle) epropnp_amis_forward deals an object's
                                 point tiles to several workgroups, which exchange partial costs through this buffer; it
                                 takes the split only if the buffer holds epropnp_amis_forward_split_bytes() bytes.  The
                                 library fills it on the stream before the launch; contents are undefined afterwards.     */
  uint64_t split_scratch_bytes;
  uint64_t* advance;          /* optional (NULL: none): `advance_count` consecutive DEVICE uint64 counters that the launch increments
                                 by one when its LAST workgroup retires -- every workgroup has read `offset_dev` long before.  With
                                 `offset_dev` (and epropnp_mc_params.rslm_offset_dev) among them a captured step advances its Philox
                                 counters inside the sampler's own launch: no add kernel per step (round 5; ~2.5 us of a replayed
                                 step, ~8 us of an eager one).  Needs `advance_ticket`.                                          */
  int32_t* advance_ticket;    /* DEVICE int32, zero before the first launch; the library returns it to zero                     */
  int32_t advance_count;
} epropnp_amis_params;

/* Bytes of `split_scratch` with which epropnp_amis_forward would split the objects of this problem over workgroups
 * (0: it would not -- enough objects to fill the device, too few point tiles, or a shape the split does not serve). */
uint64_t epropnp_amis_forward_split_bytes(const epropnp_problem* prob, int32_t mc_samples, int32_t num_iter);

/* Everything EProPnPBase.monte_carlo_forward does between its arguments and its return tuple
 * (epropnp/epropnp.py:87-196): one host call that enqueues, in order,
 *   [pnp_normalize: centre x3d, shift pose_init (common.py:103-124)] -> cost of pose_init (:121-124) ->
 *   [RSLM initialiser, and per object the cheaper of {pose_init, RSLM pose} (levenberg_marquardt.py:115-130)] ->
 *   LM solve with covariance (:132-190) -> AMIS loop (:132-182) -> [pnp_denormalize of pose_opt and the samples]. */
typedef struct epropnp_mc_params {
  epropnp_lm_params lm;          /* the main solve                                                                   */
  epropnp_amis_params amis;
  int32_t normalize;             /* 1: EProPnPBase(normalize=True)                                                   */
  int32_t init_mode;             /* 0: start LM from pose_init; 1: RSLM only (pose_init NULL);
                                    2: force_init_solve=True with pose_init: per object the cheaper of the two      */
  epropnp_lm_params rslm_lm;     /* RSLMSolver's own LM parameters (sub-problems)                                   */
  int32_t rslm_points, rslm_proposals;
  uint64_t rslm_seed, rslm_offset;
  const uint64_t* rslm_offset_dev;
  const int64_t* rslm_inds;      /* injected sub-sample indices (P,B,n) or NULL                                      */
  const float* rslm_rot;         /* injected initial rotations or NULL                                               */
  void* rslm_scratch;            /* optional scratch of epropnp_rslm_solve (see there) or NULL                       */
  uint64_t rslm_scratch_bytes;
  void* lm_scratch;              /* optional split scratch of the main epropnp_lm_solve (see there) or NULL          */
  uint64_t lm_scratch_bytes;
} epropnp_mc_params;

/*   pose_init (B,pose_len) or NULL (init_mode 1); noise as in epropnp_amis_forward
 *   scratch the caller owns (it must outlive the backward): x3d_centered (B,N,3) + offset (B,3) + pose_init_n
 *     (B,pose_len) when normalize, NULL otherwise; start_pose (B,pose_len) + start_cost (B,) when init_mode != 0
 *   -> in the (normalised) solver frame: pose_opt_n (B,pose_len), pose_cov (B,dof,dof), cost (B,) or NULL,
 *      pose_samples_n (S,B,pose_len), logweights (S,B), cost_init (B,) (NULL iff pose_init NULL)
 *   -> in the caller's frame (only when normalize; otherwise pass NULL and use the _n buffers):
 *      pose_opt (B,pose_len), pose_samples (S,B,pose_len).
 * The backward is epropnp_amis_backward on the problem with x3d = x3d_centered and pose_samples_n / pose_init_n. */
int epropnp_monte_carlo_forward(const epropnp_problem* prob, const epropnp_mc_params* par, const float* pose_init,
                                const float* noise, float* x3d_centered, float* offset, float* pose_init_n,
                                float* start_pose, float* start_cost, float* pose_opt_n, float* pose_cov, float* cost,
                                float* pose_samples_n, float* logweights, float* cost_init, float* pose_opt,
                                float* pose_samples, void* stream);

int epropnp_abi_version(void);
const char* epropnp_last_error(void);

/* Optional per-stage timing: while enabled, every kernel stage this library launches -- also the ones inside
 * epropnp_monte_carlo_forward -- is bracketed by two HIP events on its launch stream.  epropnp_profile_read synchronises
 * on the recorded events of `stage` ("evaluate_cost", "normal_equations", "lm_solve", "rslm_solve", "amis_forward",
 * "amis_backward", "adaptive_delta", "mc_loss_forward", "mc_loss_backward", "gn_step_forward", "gn_step_backward",
 * "center_points", "shift_poses") and returns their mean duration and count; bench.py's per-kernel times and roofline
 * figures come from here.  Not for use inside a hipGraph capture. */
int epropnp_profile_enable(int on);
int epropnp_profile_reset(void);
int epropnp_profile_read(const char* stage, float* mean_ms, int32_t* count);

/* The default status word.  With `epropnp_problem.status == NULL` kernels report the events above into a per-device
 * int32[2] in host memory mapped into the device: no cost without an event, and the host notices a failure with a plain
 * load -- no synchronisation.  `epropnp_async_status` returns the flags seen so far on the current device (and the first
 * offending object in flags_and_first[1]), optionally clearing them; `epropnp_async_status_word` returns the word itself
 * for callers that poll it directly (NULL when switched off with EPROPNP_ASYNC_STATUS=0).  The Python layer polls on entry
 * to every call and reports EPROPNP_ST_LM_NOT_SPD / _NONFINITE_POSE -- asynchronously: at the first call after the failing
 * kernel has run, as HIP reports its own faults -- as a RuntimeWarning, or as the RuntimeError of
 * levenberg_marquardt.py:15-19,178-181 with EPROPNP_ASYNC_STATUS=raise (the reference's LU raises only on an exactly zero
 * pivot and otherwise hands NaN poses on, which its losses zero: monte_carlo_pose_loss.py:31). */
int epropnp_async_status(int32_t* flags_and_first, int clear);
int32_t* epropnp_async_status_word(void);

/* Floats per (iteration, sample, object) of an injected-noise buffer for the given dof (8 for 6-DoF:
 * [z0,z1,z2, chi2, g0..g3]; 4 + 3*16 for 4-DoF: [z0,z1,z2, chi2, u | 16x(u1,u2,u3)]). */
int epropnp_noise_stride(int dof);

/* evaluate_pnp(..., out_cost=True) for P poses per object: the cost-only path
 * epropnp/common.py:67-100 -> camera.py:21-30,81-93 (project_b + clamp) -> cost_fun.py:45-61 (Huber).
 *   poses (P,B,pose_len)  ->  cost (P,B) */
int epropnp_evaluate_cost(const epropnp_problem* prob, const float* poses, int32_t num_poses, float* cost,
                          void* stream);

/* Gradients of sum_j a_j cost(pose_j) w.r.t. the camera intrinsics and, for one of the poses, the reduction behind the
 * gradient w.r.t. that pose -- what autograd records in the reference for cost_init = evaluate_pnp(pose=pose_init,
 * out_cost=True) and for the AMIS log-weights w.r.t. camera.cam_mats and pose_init (epropnp/epropnp.py:121-124,139-169 ->
 * common.py:90-99 -> camera.py:21-30,81-93 -> cost_fun.py:8-12,45-61).  With h = K (R X + t), g_h = d cost / d h:
 *   poses (P,B,pose_len), weights (P,B) | NULL (= 1), m_pose in [-1, P)
 *   -> grad_cam (B,3,3) | NULL  = sum_j a_j sum_n g_h (R_j X_n + t_j)^T
 *   -> grad_h_outer (B,3,4) | NULL = a_m sum_n g_h (X_n, Y_n, Z_n, 1)^T for pose m = m_pose (zeros for m_pose = -1):
 *      d/dt_m = K^T M[:,3],  d/dR_m = K^T M[:,:3]  (per object, on the caller's side). */
int epropnp_cost_pose_cam_grad(const epropnp_problem* prob, const float* poses, const float* weights, int32_t num_poses,
                               int32_t m_pose, float* grad_h_outer, float* grad_cam, void* stream);

/* evaluate_pnp(..., out_jacobian, out_residual, out_cost) fused with the normal equations the LM solver forms
 * from them: common.py:67-100 -> camera.py:10-18,81-143 (project_a, Jacobian, clip_jac) -> cost_fun.py:63-84
 * (robust rescaling) -> levenberg_marquardt.py:205-214 (JtJ, Jtr).  The (B,2N,dof) Jacobian is never written.
 *   pose (B,pose_len) -> jtj (B,dof,dof), jtr (B,dof), cost (B,) */
int epropnp_normal_equations(const epropnp_problem* prob, const float* pose, int32_t clip_jac, float* jtj,
                             float* jtr, float* cost, void* stream);

/* LMSolver.solve with a given starting pose (epropnp/levenberg_marquardt.py:132-190, _lm_iter :192-241,
 * pose_add :255-265): the whole iteration runs inside one kernel, points stay in registers.
 *   pose_init (B,pose_len) -> pose_opt (B,pose_len); pose_cov (B,dof,dof) or NULL; cost (B,) or NULL;
 *   accept_mask (B,) int32 or NULL: bit i = step i accepted (diagnostics for the trust-region flip rate). */
int epropnp_lm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, const float* pose_init, float* pose_opt,
                     float* pose_cov, float* cost, int32_t* accept_mask, void* split_scratch, uint64_t split_scratch_bytes,
                     void* stream);
/* split_scratch: optional DEVICE buffer (or NULL / 0).  With few objects of many points (LineMOD: 32 crops x 4096 dense
 * correspondences on 256 CUs) and epropnp_lm_solve_split_bytes() bytes of scratch, the points of an object are dealt to up to
 * 8 workgroups that exchange their partial normal equations through it after every sweep; the library fills it on the stream
 * before the launch, contents are undefined afterwards.  0: the library would not split this problem. */
uint64_t epropnp_lm_solve_split_bytes(const epropnp_problem* prob, const epropnp_lm_params* lm);

/* The AMIS loop of EProPnPBase.monte_carlo_forward (epropnp/epropnp.py:132-182) including initial_fit,
 * gen_new_distr/gen_old_distr, estimate_params (:199-342), the proposal densities (epropnp/distributions.py,
 * pyro MultivariateStudentT) and the weight algebra (:156-169).
 *   pose_opt (B,pose_len), pose_cov (B,dof,dof) from the solver
 *   noise: NULL (on-device Philox) or (B,K,S/K,epropnp_noise_stride(dof)) injected base draws
 *   -> pose_samples (S,B,pose_len), logweights (S,B)
 *   -> proposals (optional, may be NULL): (B,K,40) fitted proposal parameters, for diagnostics/tests.
 * Arithmetic: fp32.  The pose x point projection of the cost sweep (camera.py:21-30) runs on v_mfma_f32_16x16x32_bf16 with each
 * fp32 operand split into three bf16 pieces that sum to it exactly (8 of the 9 cross products, fp32 accumulation: 1.4e-7
 * relative, the fp32 MFMA's 1.2e-7) wherever the points are register-resident; environment EPROPNP_FWD_PROJ=f32 selects
 * v_mfma_f32_16x16x4_f32 (and EPROPNP_BWD_PROJ=f32 does so for epropnp_amis_backward). */
int epropnp_amis_forward(const epropnp_problem* prob, const epropnp_amis_params* amis, const float* pose_opt,
                         const float* pose_cov, const float* noise, float* pose_samples, float* logweights,
                         float* proposals, void* stream);

/* Backward of  sum_b [ g_init[b]*cost(pose_init[b]) + sum_j g_logw[j,b]*logw[j,b] ]  w.r.t. x3d, x2d, w2d, delta:
 * what autograd replays through evaluate_pnp in the reference, recomputed
 * from the points instead of stored activations.  logw = -cost - const  =>  weight of sample j is -g_logw[j,b].
 * Samples whose TOTAL |weight| is below 2^-24 of the object's total (one fp32 rounding of the sum) are skipped;
 * environment EPROPNP_BWD_DROP=<fraction> changes that budget, 0 evaluates every non-zero sample (exact).
 *   pose_samples (S,B,pose_len), grad_logweights (S,B), pose_init (B,pose_len) or NULL, grad_cost_init (B,) or NULL
 *   -> grad_x3d (B,N,3), grad_x2d (B,N,2), grad_w2d (B,N,2), grad_delta (B,) */
int epropnp_amis_backward(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                          int32_t mc_samples, const float* pose_init, const float* grad_cost_init,
                          float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta, void* stream);

/* The same backward for FEW objects (a single CU is busy ~70 us with one object's 512 x 512 point-poses): the point
 * chunks of an object are dealt to `num_split` workgroups.  grad_x3d / grad_x2d / grad_w2d are bit-identical to
 * epropnp_amis_backward; grad_delta comes back as partials
 *   grad_delta_parts (B, num_split)   with   grad_delta[b] = sum_c grad_delta_parts[b, c]
 * for the caller to add in a fixed order (no atomics).  1 <= num_split <= min(16, ceil(num_pts / 64)); needs the
 * LDS-resident pose table (mc_samples <~ 2700), EPROPNP_EINVAL otherwise. */
int epropnp_amis_backward_split(const epropnp_problem* prob, const float* pose_samples, const float* grad_logweights,
                                int32_t mc_samples, const float* pose_init, const float* grad_cost_init, int32_t num_split,
                                float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta_parts, void* stream);

/* AdaptiveHuberPnPCost.set_param (epropnp/cost_fun.py:123-126):
 *   delta[b] = mean(w2d[b]) * sqrt(sum_xy var_N(x2d[b])) * relative_delta      (unbiased variance)
 * x2d (B,N,2), w2d (B,N,2) -> delta (B,), stats (B,4) = [mean_w, x2d_std, mean_x, mean_y] (kept for the backward,
 * which is three broadcast expressions evaluated by the caller). */
int epropnp_adaptive_delta(const float* x2d, const float* w2d, int32_t num_obj, int32_t num_pts, float relative_delta,
                           float* delta, float* stats, void* stream);

/* Monte-Carlo pose loss per object (EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py:28-32):
 *   loss[b] = cost_target[b] + logsumexp_j logweights[j,b];  NaN -> 0.   lse (B,) is kept for the backward.
 * logweights (S,B), cost_target (B,) or NULL. */
int epropnp_mc_loss_forward(const float* logweights, const float* cost_target, int32_t mc_samples, int32_t num_obj,
                            float* loss, float* lse, void* stream);
/* grad_logweights[j,b] = grad_loss[b] * exp(logweights[j,b] - lse[b])   (0 where the loss was NaN);
 * grad_cost_target (B,) or NULL: grad_loss[b] (0 where the loss was NaN). */
int epropnp_mc_loss_backward(const float* logweights, const float* lse, const float* loss, const float* grad_loss,
                             int32_t mc_samples, int32_t num_obj, float* grad_logweights, float* grad_cost_target,
                             void* stream);
/* The SCALAR both reference loss modules return, from the per-object losses of epropnp_mc_loss_forward, in one
 * single-workgroup launch (EPro-PnP-Det epropnp_det/models/losses/monte_carlo_pose_loss.py:41-66 incl. mmdet's
 * weight_reduce_loss; EPro-PnP-6DoF lib/models/monte_carlo_pose_loss.py:20-35):
 *   norm_factor[0] <- (1 - momentum) * norm_factor[0] + momentum * norm_factor_in[0]     if norm_factor_in != NULL (training)
 *   out[0] = (sum_b weight[b] * loss[b]) * scale / norm_factor[0],    out[1] = scale / norm_factor[0]  (for the backward)
 * scale = loss_weight / num_obj ('mean'), loss_weight ('sum') or loss_weight / avg_factor.  weight (B,) or NULL (ones),
 * norm_factor (1,) device scalar or NULL (1.0), out (2,).  Fixed summation order: bit-reproducible.
 *   norm_factor_in: `norm_factor_in_count` values `norm_factor_in_stride` floats apart, averaged in index order -- the
 *   world mean of mmdet's reduce_mean read straight out of the receive buffer of the step's one all-gather (count = ranks,
 *   stride = floats per rank; sharding.ObjectExchange); count 1 for a plain scalar. */
int epropnp_mc_loss_reduce(const float* loss, const float* weight, int32_t num_obj, float scale, float momentum,
                           const float* norm_factor_in, int32_t norm_factor_in_count, int64_t norm_factor_in_stride,
                           float* norm_factor, float* out, void* stream);

/* Send buffer of the Det step's one collective (sharding.ObjectExchange), packed by ONE launch:
 *   send[0 .. n_scalars)            = scalars[i]                       (or, with sum_src != NULL, send[0] = sum_scale * sum(sum_src[0 .. sum_floats)),
 *                                                                      a fixed-order sum -- the detection head's norm_factor input,
 *                                                                      deform_pnp_head.py:870 -- and scalars[i] for i >= 1)
 *   send[n_scalars .. + row_floats) = rows                             (this rank's per-object outputs, flattened)
 * scalars may be NULL when n_scalars == 0 or (n_scalars == 1 and sum_src != NULL).
 * sum_row_weight (or NULL): sum_src is a (rows, sum_row_len) array and element [i, c] counts sum_row_weight[i] times -- the head's
 * `(scale * sample_weights[:, None]).sum() / max(2 n, 1)` in one go. */
int epropnp_exchange_pack(const float* rows, uint64_t row_floats, const float* scalars, int32_t n_scalars,
                          const float* sum_src, uint64_t sum_floats, float sum_scale, const float* sum_row_weight,
                          int32_t sum_row_len, float* send, void* stream);
/* Its backward through epropnp_mc_loss_forward: with g_b = grad_out[0] * coef[0] * weight[b] (coef = out + 1 of the forward),
 *   grad_logweights[j,b] = g_b * exp(logweights[j,b] - lse[b]),  grad_cost_target[b] = g_b (or NULL)   (0 where the loss was NaN).
 * grad_out and coef are device scalars: nothing is read back, the node is capturable. */
int epropnp_mc_loss_reduce_backward(const float* logweights, const float* lse, const float* weight, const float* coef,
                                    const float* grad_out, int32_t mc_samples, int32_t num_obj, float* grad_logweights,
                                    float* grad_cost_target, void* stream);

/* LMSolver.gn_step (epropnp/levenberg_marquardt.py:243-253), the differentiable Gauss-Newton step behind
 * `pose_opt_plus`:  step = -(J^T J + eps I)^-1 J^T r  at `pose` (clip_jac on).   pose (B,pose_len) -> step (B,dof). */
int epropnp_gn_step_forward(const epropnp_problem* prob, float eps, const float* pose, float* step, void* stream);
/* Its backward w.r.t. x3d, x2d, w2d, delta (the pose is not differentiated, as in the reference):
 * grad_step (B,dof) -> grad_x3d (B,N,3), grad_x2d (B,N,2), grad_w2d (B,N,2), grad_delta (B,). */
int epropnp_gn_step_backward(const epropnp_problem* prob, float eps, const float* pose, const float* grad_step,
                             float* grad_x3d, float* grad_x2d, float* grad_w2d, float* grad_delta, void* stream);

/* The same step composed with LMSolver.pose_add (levenberg_marquardt.py:70-72,255-265):
 *   pose_plus = pose (+) gn_step(pose)   (B,pose_len)   -- `pose_opt_plus` of LMSolver.forward in one launch,
 * and its backward from grad_pose_plus (B,pose_len) (the adjoint of pose_add is applied inside the kernel). */
int epropnp_pose_opt_plus_forward(const epropnp_problem* prob, float eps, const float* pose, float* pose_plus,
                                  void* stream);
int epropnp_pose_opt_plus_backward(const epropnp_problem* prob, float eps, const float* pose,
                                   const float* grad_pose_plus, float* grad_x3d, float* grad_x2d, float* grad_w2d,
                                   float* grad_delta, void* stream);

/* Sub-sample indices of the RSLM initialiser (epropnp/levenberg_marquardt.py:305-308): for each of the P x B
 * (proposal, object) rows draw n_pts distinct point indices with probability proportional to mean(w2d[b,n,:]),
 * sequentially without replacement (exponential-race keys -log(u)/w, the n_pts smallest win; same law as
 * torch.multinomial(replacement=False)).   w2d (B,N,2) -> inds (P,B,n_pts) int64, Philox(seed, offset). */
int epropnp_rslm_draw(const float* w2d, int32_t num_obj, int32_t num_pts, int32_t num_proposals, int32_t n_pts,
                      uint64_t seed, uint64_t offset, int64_t* inds, void* stream);

/* pnp_normalize (epropnp/common.py:103-124): offset[b] = mean_n x3d[b,n,:], x3d_centered = x3d - offset.
 * x3d (B,N,3) -> offset (B,3), x3d_centered (B,N,3). */
int epropnp_center_points(const float* x3d, int32_t num_obj, int32_t num_pts, float* offset, float* x3d_centered,
                          void* stream);
/* The pose half of pnp_normalize / pnp_denormalize (epropnp/common.py:118-136):
 * out[j,b] = pose[j,b] with translation += sign * R(pose[j,b]) offset[b]   (sign +1 normalise, -1 denormalise).
 * pose (P,B,pose_len), offset (B,3) -> out (P,B,pose_len); out may alias pose. */
int epropnp_shift_poses(const float* pose, const float* offset, int32_t num_poses, int32_t num_obj, int32_t dof,
                        float sign, float* out, void* stream);

/* Correspondence pre-processing of the reference's training loops, fused (the callers' side of the layer):
 *   x3d = noc * dim                                            EPro-PnP-6DoF/lib/train.py:141, Det deform_pnp_head.py:873
 *   mode 0: w2d = softmax_N(logits) * scale                    Det deform_pnp_head.py:418-423,874
 *   mode 1: w2d = exp(logits - mean_N(logits) - log N) * scale EPro-PnP-6DoF/lib/train.py:163-166
 * noc (B,N,3), dim (B,3) (both NULL with x3d NULL: weights only), logits (B,N,2), scale (B,2) or NULL
 * -> x3d (B,N,3), w2d (B,N,2), stats (B,4) kept for the backward. */
#define EPROPNP_W2D_SOFTMAX 0
#define EPROPNP_W2D_MEAN_EXP 1
int epropnp_prepare_forward(const float* noc, const float* dim, const float* logits, const float* scale,
                            int32_t num_obj, int32_t num_pts, int32_t mode, float* x3d, float* w2d, float* stats,
                            void* stream);
/* grad_x3d (B,N,3) or NULL, grad_w2d (B,N,2) -> grad_noc (B,N,3), grad_dim (B,3), grad_logits (B,N,2),
 * grad_scale (B,2) (grad_noc / grad_dim / grad_scale may be NULL). */
int epropnp_prepare_backward(const float* noc, const float* dim, const float* logits, const float* scale,
                             const float* stats, const float* grad_x3d, const float* grad_w2d, int32_t num_obj,
                             int32_t num_pts, int32_t mode, float* grad_noc, float* grad_dim, float* grad_logits,
                             float* grad_scale, void* stream);

/* The same pre-processing reading the network's DENSE maps (EPro-PnP-6DoF/lib/train.py:141-166): the maps are gathered
 * at `inds` (pixel index row * width + col; the reference draws them with np.random.choice, :157-162:
 * `x.flatten(2).transpose(-1, -2)[batch_inds, sample_inds]`) without materialising the transposed maps, and the pixel
 * grid of the sampled pixels is generated in place of the meshgrid of :147-152:
 *   x2d[b,n] = (box[b,0] + col * box[b,2], box[b,1] + row * box[b,2])     box = [wh_begin_x, wh_begin_y, wh_unit]
 * noc_map (B,3,H,W), dim (B,3) (both NULL with x3d NULL), logit_map (B,2,H,W), scale (B,2) or NULL, box (B,3) (NULL with
 * x2d NULL), inds (B,N) int64 -> x3d (B,N,3), x2d (B,N,2), w2d (B,N,2), stats (B,4). */
int epropnp_prepare_dense_forward(const float* noc_map, const float* dim, const float* logit_map, const float* scale,
                                  const float* box, const int64_t* inds, int32_t num_obj, int32_t num_pts, int32_t height,
                                  int32_t width, int32_t mode, float* x3d, float* x2d, float* w2d, float* stats,
                                  void* stream);
/* grad_x3d (B,N,3) or NULL, grad_w2d (B,N,2) -> grad_noc_map (B,3,H,W), grad_logit_map (B,2,H,W) (zero-filled here, then
 * scattered at `inds`; repeated indices accumulate), grad_dim (B,3), grad_scale (B,2) or NULL. */
int epropnp_prepare_dense_backward(const float* noc_map, const float* dim, const float* logit_map, const float* scale,
                                   const int64_t* inds, const float* stats, const float* grad_x3d, const float* grad_w2d,
                                   int32_t num_obj, int32_t num_pts, int32_t height, int32_t width, int32_t mode,
                                   float* grad_noc_map, float* grad_dim, float* grad_logit_map, float* grad_scale,
                                   void* stream);

/* Backward of epropnp_shift_poses w.r.t. the pose (the offset is a constant, pnp_normalize detaches it):
 * grad_out (P,B,pose_len) -> grad_pose (P,B,pose_len). */
int epropnp_shift_poses_backward(const float* pose, const float* offset, const float* grad_out, int32_t num_poses,
                                 int32_t num_obj, int32_t dof, float sign, float* grad_pose, void* stream);

/* RSLMSolver.solve (epropnp/levenberg_marquardt.py:283-353) in one launch: center_based_init, weighted sub-sampling
 * of `num_points` (<= 16) correspondences per proposal, random initial rotations, `num_proposals` LM/GN solves per
 * object on the sub-samples (parameters `lm`, as LMSolver.solve), full-set cost of every proposal, argmin.
 *   inds: NULL (drawn on the device, same stream as epropnp_rslm_draw) or (P,B,num_points) int64 injected indices
 *   rot:  NULL (drawn on the device) or (P,B,1) yaw / (P,B,4) unit quaternions, injected
 *   offset_dev: optional device counter added to `offset` at run time (hipGraph replay), or NULL
 *   scratch: optional DEVICE buffer of scratch_bytes bytes (or NULL / 0): with epropnp_rslm_solve_scratch_bytes() bytes the
 *            proposals of an object are dealt to 2 or 4 workgroups (finer load balance: 600 objects x 64 proposals 66 ->
 *            51 us) whose candidates meet there; contents undefined afterwards.  Results do not depend on it.
 *   -> pose (B,pose_len) best proposal, cost (B,) its full-set Huber cost (may be NULL).   num_pts in [2, 512]. */
int epropnp_rslm_solve(const epropnp_problem* prob, const epropnp_lm_params* lm, int32_t num_proposals,
                       int32_t num_points, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                       const int64_t* inds, const float* rot, float* pose, float* cost, void* scratch,
                       uint64_t scratch_bytes, void* stream);
uint64_t epropnp_rslm_solve_scratch_bytes(const epropnp_problem* prob, int32_t num_proposals);

#ifdef __cplusplus
}
#endif
#endif /* EPROPNP_HIP_H */
