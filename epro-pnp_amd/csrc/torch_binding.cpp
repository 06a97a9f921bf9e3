// torch_binding.cpp -- the autograd nodes of the EPro-PnP layer as C++ torch::autograd::Function's over the C ABI
// (include/epropnp_hip.h).  Host code only: compiled with g++ against the torch headers and linked to
// libepropnp_hip.so; every kernel lives in that library.
//
// Why it exists: at the launch-bound shapes (EPro-PnP-Det: 600 objects x 128 points; LineMOD training: 32 x 512) a step
// is ~10 kernel stages of a few microseconds each, and the Python autograd nodes (set_param, layer, loss) cost more
// host time than the GPU needs -- most of it in the BACKWARD, where the engine has to re-enter the interpreter once per
// node.  These nodes run their backward without the interpreter.  epropnp/functional.py uses them when the module
// is importable (built by epro-pnp_amd/build.py next to the HIP library) and the ctypes nodes otherwise; both call the
// same entry points of the same library (bit-identical results, tests/test_torch_binding.py).
//
// Reference spans: AdaptiveHuberPnPCost.set_param (epropnp/cost_fun.py:123-126), EProPnPBase.monte_carlo_forward
// (epropnp/epropnp.py:87-196) + autograd replay of evaluate_pnp, MonteCarloPoseLoss.forward
// (EPro-PnP-6DoF/lib/models/monte_carlo_pose_loss.py:28-32).
#include <torch/extension.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/epropnp_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using OptTensor = c10::optional<Tensor>;

inline float* fptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline float* fptr(const OptTensor& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }

void check(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", epropnp_last_error());
}

// ---------------------------------------------------------------------------------------------------------------------
// delta_b = mean(w2d_b) * sqrt(sum_xy var_N(x2d_b)) * relative_delta
struct AdaptiveDelta : public torch::autograd::Function<AdaptiveDelta> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x2d, const Tensor& w2d, double rel, int64_t stream) {
    const Tensor x = x2d.detach().contiguous(), w = w2d.detach().contiguous();
    const int64_t B = x.size(0), N = x.size(1);
    Tensor delta = torch::empty({B}, x.options()), stats = torch::empty({B, 4}, x.options());
    check(epropnp_adaptive_delta(fptr(x), fptr(w), (int32_t)B, (int32_t)N, (float)rel, fptr(delta), fptr(stats),
                                 (void*)stream), "epropnp_adaptive_delta");
    ctx->save_for_backward({x, stats});
    ctx->saved_data["rel"] = rel;
    ctx->saved_data["N"] = N;
    ctx->set_materialize_grads(false);
    ctx->mark_non_differentiable({stats});      // handed to the layer: epropnp_problem.delta_stats
    return {delta, stats};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const Tensor g = grads[0];
    if (!g.defined()) return {Tensor(), Tensor(), Tensor(), Tensor()};
    const auto saved = ctx->get_saved_variables();
    const Tensor& x = saved[0];
    const Tensor& stats = saved[1];
    const double rel = ctx->saved_data["rel"].toDouble();
    const int64_t N = ctx->saved_data["N"].toInt();
    const Tensor mw = stats.select(1, 0), sd = stats.select(1, 1);
    Tensor gx, gw;
    if (ctx->needs_input_grad(1))      // d delta / d w = std * rel / (2N), the same for every element of the object
      gw = (g * sd * (rel / (2.0 * (double)N))).view({-1, 1, 1}).expand({-1, N, 2});
    if (ctx->needs_input_grad(0)) {    // d delta / d x = mean_w * rel * (x - mean) / ((N-1) std)
      const Tensor coef = g * mw * rel / ((double)(N - 1) * sd.clamp_min(1e-30));
      gx = coef.view({-1, 1, 1}) * (x - stats.slice(1, 2, 4).unsqueeze(1));
    }
    return {gx, gw, Tensor(), Tensor()};
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// per-object Monte-Carlo pose loss: cost_target + logsumexp_S(logweights), NaN -> 0
struct McPoseLoss : public torch::autograd::Function<McPoseLoss> {
  static Tensor forward(AutogradContext* ctx, const Tensor& logw, const OptTensor& cost_target, int64_t stream) {
    const Tensor lw = logw.detach().contiguous();
    const int64_t S = lw.size(0), B = lw.size(1);
    Tensor ct;
    if (cost_target.has_value() && cost_target->defined()) ct = cost_target->detach().contiguous();
    Tensor loss = torch::empty({B}, lw.options()), lse = torch::empty({B}, lw.options());
    check(epropnp_mc_loss_forward(fptr(lw), fptr(ct), (int32_t)S, (int32_t)B, fptr(loss), fptr(lse), (void*)stream),
          "epropnp_mc_loss_forward");
    ctx->save_for_backward({lw, lse, loss});
    ctx->saved_data["stream"] = stream;
    ctx->saved_data["has_cost_target"] = ct.defined();
    ctx->set_materialize_grads(false);
    return loss;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    if (!grads[0].defined()) return {Tensor(), Tensor(), Tensor()};
    const auto saved = ctx->get_saved_variables();
    const Tensor &lw = saved[0], &lse = saved[1], &loss = saved[2];
    const Tensor g = grads[0].contiguous();
    const int64_t S = lw.size(0), B = lw.size(1);
    Tensor glw = torch::empty_like(lw), gct;
    // needs_input_grad indexes the edges of the tensor inputs actually passed: without a cost_target there is no edge 1
    if (ctx->saved_data["has_cost_target"].toBool() && ctx->needs_input_grad(1)) gct = torch::empty_like(g);
    check(epropnp_mc_loss_backward(fptr(lw), fptr(lse), fptr(loss), fptr(g), (int32_t)S, (int32_t)B, fptr(glw), fptr(gct),
                                   (void*)ctx->saved_data["stream"].toInt()), "epropnp_mc_loss_backward");
    return {glw, gct, Tensor()};
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// the REDUCED Monte-Carlo pose loss of the two reference loss modules (include/epropnp_hip.h: epropnp_mc_loss_reduce):
// (sum_b weight_b loss_b) * scale / norm_factor as a 0-dim tensor; norm_factor's running estimate is updated in place
struct McPoseLossReduced : public torch::autograd::Function<McPoseLossReduced> {
  static Tensor forward(AutogradContext* ctx, const Tensor& logw, const OptTensor& cost_target, const OptTensor& weight,
                        double scale, double momentum, const OptTensor& nf_in, const OptTensor& norm_factor, int64_t stream) {
    const Tensor lw = logw.detach().contiguous();
    const int64_t S = lw.size(0), B = lw.size(1);
    Tensor ct, w;
    if (cost_target.has_value() && cost_target->defined()) ct = cost_target->detach().contiguous();
    if (weight.has_value() && weight->defined()) w = weight->detach().contiguous();
    Tensor loss = torch::empty({B}, lw.options()), lse = torch::empty({B}, lw.options()), out = torch::empty({2}, lw.options());
    check(epropnp_mc_loss_forward(fptr(lw), fptr(ct), (int32_t)S, (int32_t)B, fptr(loss), fptr(lse), (void*)stream),
          "epropnp_mc_loss_forward");
    // nf_in: a scalar, or the (ranks,) STRIDED view of the exchange's receive buffer whose mean the kernel takes itself
    int32_t nf_count = 1;
    int64_t nf_stride = 1;
    if (nf_in.has_value() && nf_in->defined() && nf_in->dim() == 1 && nf_in->numel() > 1) {
      nf_count = (int32_t)nf_in->numel();
      nf_stride = nf_in->stride(0);
    }
    check(epropnp_mc_loss_reduce(fptr(loss), fptr(w), (int32_t)B, (float)scale, (float)momentum, fptr(nf_in), nf_count, nf_stride,
                                 fptr(norm_factor), fptr(out), (void*)stream), "epropnp_mc_loss_reduce");
    ctx->save_for_backward({lw, lse, out, w});
    ctx->saved_data["stream"] = stream;
    ctx->saved_data["has_cost_target"] = ct.defined();
    ctx->set_materialize_grads(false);
    return out.select(0, 0);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    if (!grads[0].defined()) return variable_list(8);
    const auto saved = ctx->get_saved_variables();
    const Tensor &lw = saved[0], &lse = saved[1], &out = saved[2], &w = saved[3];
    const Tensor g = grads[0].to(torch::kFloat32).reshape({1}).contiguous();
    const int64_t S = lw.size(0), B = lw.size(1);
    Tensor glw = torch::empty_like(lw), gct;
    if (ctx->saved_data["has_cost_target"].toBool() && ctx->needs_input_grad(1)) gct = torch::empty({B}, lw.options());
    check(epropnp_mc_loss_reduce_backward(fptr(lw), fptr(lse), fptr(w), fptr(out) + 1, fptr(g), (int32_t)S, (int32_t)B,
                                          fptr(glw), fptr(gct), (void*)ctx->saved_data["stream"].toInt()),
          "epropnp_mc_loss_reduce_backward");
    variable_list r(8);
    r[0] = glw;
    r[1] = gct;
    return r;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// monte_carlo_forward: (x3d, x2d, w2d, delta) -> pose_opt_n, samples_n, logweights [diff], cost, cost_init [diff],
//                                               pose_opt, samples, x3d_centered, offset
struct ProblemTensors {      // contiguous fp32 device tensors behind an epropnp_problem
  Tensor x3d, x2d, w2d, cam, lb, ub, delta, status, dstats;
  double z_min = 0.1, huber_eps = 1e-10, drel = 0.0;
  int64_t dof = 6;
  epropnp_problem c() const {
    epropnp_problem p = {};
    p.x3d = fptr(x3d); p.x2d = fptr(x2d); p.w2d = fptr(w2d); p.cam_mats = fptr(cam);
    p.lb = fptr(lb); p.ub = fptr(ub); p.delta = fptr(delta);
    p.z_min = (float)z_min;
    p.num_obj = (int32_t)x2d.size(0); p.num_pts = (int32_t)x2d.size(1); p.dof = (int32_t)dof;
    p.huber_eps = (float)huber_eps;
    p.status = status.defined() ? status.data_ptr<int32_t>() : nullptr;
    p.delta_stats = fptr(dstats); p.delta_relative = (float)drel;
    return p;
  }
};

int backward_launch(const epropnp_problem& q, const Tensor& samples, const Tensor& glw, const Tensor& pin, const Tensor& gin,
                    int64_t nsplit, Tensor& gx3d, Tensor& gx2d, Tensor& gw2d, Tensor& gdel, int64_t stream) {
  const int64_t B = q.num_obj, N = q.num_pts, S = samples.size(0);
  gx3d = torch::empty({B, N, 3}, samples.options());
  gx2d = torch::empty({B, N, 2}, samples.options());
  gw2d = torch::empty({B, N, 2}, samples.options());
  if (nsplit > 1) {
    Tensor parts = torch::empty({B, nsplit}, samples.options());
    const int rc = epropnp_amis_backward_split(&q, fptr(samples), fptr(glw), (int32_t)S, fptr(pin), fptr(gin), (int32_t)nsplit,
                                               fptr(gx3d), fptr(gx2d), fptr(gw2d), fptr(parts), (void*)stream);
    // (the per-workgroup partials are only added up for a caller that wants d/d delta: with the threshold's gradient folded into
    // grad_w2d -- q.delta_stats -- nobody does, and the sum would be one more launch of a launch-bound step)
    if (q.delta_stats == nullptr) gdel = parts.sum(1);
    return rc;
  }
  gdel = torch::empty({B}, samples.options());
  return epropnp_amis_backward(&q, fptr(samples), fptr(glw), (int32_t)S, fptr(pin), fptr(gin), fptr(gx3d), fptr(gx2d),
                               fptr(gw2d), fptr(gdel), (void*)stream);
}

struct FusedMonteCarlo : public torch::autograd::Function<FusedMonteCarlo> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x3d, const Tensor& x2d, const Tensor& w2d,
                               const OptTensor& delta, const Tensor& x3d_c, const Tensor& x2d_c, const Tensor& w2d_c,
                               const Tensor& cam_c, const OptTensor& lb_c, const OptTensor& ub_c, const Tensor& delta_c,
                               const OptTensor& status, double z_min, double huber_eps, int64_t dof,
                               const OptTensor& pose_init, const OptTensor& noise, const std::string& mc_params,
                               bool with_cost, int64_t nsplit, int64_t stream, const OptTensor& delta_stats,
                               double delta_rel) {
    (void)x3d; (void)x2d; (void)w2d;      // graph inputs; the kernels read the contiguous fp32 views
    TORCH_CHECK(mc_params.size() == sizeof(epropnp_mc_params), "mc_params: ", mc_params.size(), " bytes, expected ",
                sizeof(epropnp_mc_params));
    epropnp_mc_params par;
    std::memcpy(&par, mc_params.data(), sizeof(par));
    ProblemTensors pt;
    pt.x3d = x3d_c; pt.x2d = x2d_c; pt.w2d = w2d_c; pt.cam = cam_c; pt.delta = delta_c;
    if (lb_c.has_value() && ub_c.has_value()) { pt.lb = *lb_c; pt.ub = *ub_c; }
    if (status.has_value()) pt.status = *status;
    pt.z_min = z_min; pt.huber_eps = huber_eps; pt.dof = dof;
    const epropnp_problem prob = pt.c();
    const int64_t B = prob.num_obj, N = prob.num_pts, PL = dof == 6 ? 7 : 4, d = dof, S = par.amis.mc_samples;
    const auto opt = x2d_c.options();
    Tensor pin, nz;
    if (pose_init.has_value() && pose_init->defined()) pin = pose_init->detach().contiguous();
    if (noise.has_value() && noise->defined()) nz = noise->detach().contiguous();
    const bool normalize = par.normalize != 0;
    Tensor x3d_ctr, offset, pin_n, start_pose, start_cost, cost, cost_init, pose_opt, samples;
    if (normalize) {
      x3d_ctr = torch::empty({B, N, 3}, opt);
      offset = torch::empty({B, 3}, opt);
      if (pin.defined()) pin_n = torch::empty({B, PL}, opt);
      pose_opt = torch::empty({B, PL}, opt);
      samples = torch::empty({S, B, PL}, opt);
    }
    if (par.init_mode != 0) { start_pose = torch::empty({B, PL}, opt); start_cost = torch::empty({B}, opt); }
    Tensor pose_opt_n = torch::empty({B, PL}, opt), pose_cov = torch::empty({B, d, d}, opt);
    Tensor samples_n = torch::empty({S, B, PL}, opt), logw = torch::empty({S, B}, opt);
    if (with_cost) cost = torch::empty({B}, opt);
    if (pin.defined()) cost_init = torch::empty({B}, opt);
    check(epropnp_monte_carlo_forward(&prob, &par, fptr(pin), fptr(nz), fptr(x3d_ctr), fptr(offset), fptr(pin_n),
                                      fptr(start_pose), fptr(start_cost), fptr(pose_opt_n), fptr(pose_cov), fptr(cost),
                                      fptr(samples_n), fptr(logw), fptr(cost_init), fptr(pose_opt), fptr(samples),
                                      (void*)stream), "epropnp_monte_carlo_forward");
    // ---- what the recompute backward needs: the problem in the solver frame, the samples, pose_init in that frame
    ctx->save_for_backward({samples_n});
    ctx->saved_data["x3d"] = normalize ? x3d_ctr : x3d_c;
    ctx->saved_data["x2d"] = x2d_c; ctx->saved_data["w2d"] = w2d_c; ctx->saved_data["cam"] = cam_c;
    ctx->saved_data["delta"] = delta_c;
    ctx->saved_data["lb"] = pt.lb.defined() ? c10::IValue(pt.lb) : c10::IValue();
    ctx->saved_data["ub"] = pt.ub.defined() ? c10::IValue(pt.ub) : c10::IValue();
    ctx->saved_data["pin"] = pin.defined() ? c10::IValue(normalize ? pin_n : pin) : c10::IValue();
    ctx->saved_data["z_min"] = z_min; ctx->saved_data["huber_eps"] = huber_eps; ctx->saved_data["dof"] = dof;
    ctx->saved_data["nsplit"] = nsplit; ctx->saved_data["stream"] = stream;
    ctx->saved_data["delta_dim"] = (delta.has_value() && delta->defined()) ? (int64_t)delta->dim() : (int64_t)-1;
    // delta = adaptive_delta(., w2d) of THIS w2d: the backward kernel adds grad_delta's way into grad_w2d itself
    // (epropnp_problem.delta_stats) and delta receives no gradient from this node
    const bool fold = delta_stats.has_value() && delta_stats->defined();
    ctx->saved_data["dstats"] = fold ? c10::IValue(delta_stats->detach().contiguous()) : c10::IValue();
    ctx->saved_data["drel"] = delta_rel;
    // version guard on the caller-visible inputs (the contiguous views share their version counters when no copy was made)
    std::vector<int64_t> vers;
    for (const Tensor* t : {&x3d_c, &x2d_c, &w2d_c, &delta_c, &cam_c}) vers.push_back((int64_t)t->_version());
    ctx->saved_data["guard_x3d"] = x3d_c;      // keeps the un-centred points alive for the version check
    ctx->saved_data["versions"] = vers;
    ctx->set_materialize_grads(false);
    // an autograd node cannot return undefined tensors: absent outputs travel as 0-element placeholders (B > 0 here)
    auto some = [&](const Tensor& t) { return t.defined() ? t : torch::empty({0}, opt); };
    variable_list outs = {pose_opt_n, samples_n, logw, some(cost), some(cost_init), some(pose_opt), some(samples),
                          some(x3d_ctr), some(offset)};
    variable_list nd = {outs[0], outs[1], outs[3], outs[5], outs[6], outs[7], outs[8]};
    if (!cost_init.defined()) nd.push_back(outs[4]);
    ctx->mark_non_differentiable(nd);
    return outs;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list out(23);
    const Tensor g_logw_in = grads[2], g_ci = grads[4];
    if (!g_logw_in.defined() && !g_ci.defined()) return out;
    const auto vers = ctx->saved_data["versions"].toIntVector();
    const Tensor gx3d0 = ctx->saved_data["guard_x3d"].toTensor();
    ProblemTensors pt;
    pt.x3d = ctx->saved_data["x3d"].toTensor(); pt.x2d = ctx->saved_data["x2d"].toTensor();
    pt.w2d = ctx->saved_data["w2d"].toTensor(); pt.cam = ctx->saved_data["cam"].toTensor();
    pt.delta = ctx->saved_data["delta"].toTensor();
    if (!ctx->saved_data["lb"].isNone()) { pt.lb = ctx->saved_data["lb"].toTensor(); pt.ub = ctx->saved_data["ub"].toTensor(); }
    pt.z_min = ctx->saved_data["z_min"].toDouble(); pt.huber_eps = ctx->saved_data["huber_eps"].toDouble();
    pt.dof = ctx->saved_data["dof"].toInt();
    const bool fold = !ctx->saved_data["dstats"].isNone();
    if (fold) { pt.dstats = ctx->saved_data["dstats"].toTensor(); pt.drel = ctx->saved_data["drel"].toDouble(); }
    {
      const Tensor* ts[5] = {&gx3d0, &pt.x2d, &pt.w2d, &pt.delta, &pt.cam};
      for (int i = 0; i < 5; ++i)
        TORCH_CHECK((int64_t)ts[i]->_version() == vers[i],
                    "one of the variables needed for gradient computation has been modified by an inplace operation "
                    "(EPro-PnP recomputes its backward from x3d / x2d / w2d / delta / cam_mats: they must stay unchanged "
                    "between forward and backward)");
    }
    const Tensor samples_n = ctx->get_saved_variables()[0];
    const epropnp_problem q = pt.c();
    Tensor glw = g_logw_in.defined() ? g_logw_in.contiguous() : torch::full({samples_n.size(0), samples_n.size(1)}, 0.0, samples_n.options());   // (a fill kernel; zeros() is a memset node under capture)
    Tensor pin, gin;
    if (g_ci.defined() && !ctx->saved_data["pin"].isNone()) { pin = ctx->saved_data["pin"].toTensor(); gin = g_ci.contiguous(); }
    Tensor gx3d, gx2d, gw2d, gdel;
    check(backward_launch(q, samples_n, glw, pin, gin, ctx->saved_data["nsplit"].toInt(), gx3d, gx2d, gw2d, gdel,
                          ctx->saved_data["stream"].toInt()), "epropnp_amis_backward");
    if (ctx->needs_input_grad(0)) out[0] = gx3d;
    if (ctx->needs_input_grad(1)) out[1] = gx2d;
    if (ctx->needs_input_grad(2)) out[2] = gw2d;
    const int64_t ddim = ctx->saved_data["delta_dim"].toInt();
    if (!fold && ddim >= 0 && ctx->needs_input_grad(3)) out[3] = (ddim == 0) ? gdel.sum() : gdel;
    return out;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// pose_opt_plus = pose (+) gn_step(pose)  (LMSolver.forward :70-72 -> gn_step :243-253 -> pose_add :255-265), with_plus
// false: the bare step.  Differentiable w.r.t. x3d, x2d, w2d, delta (the pose is not differentiated, as in the reference).
struct GnStep : public torch::autograd::Function<GnStep> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x3d, const Tensor& x2d, const Tensor& w2d, const OptTensor& delta,
                        const Tensor& x3d_c, const Tensor& x2d_c, const Tensor& w2d_c, const Tensor& cam_c,
                        const OptTensor& lb_c, const OptTensor& ub_c, const Tensor& delta_c, const OptTensor& status,
                        double z_min, double huber_eps, int64_t dof, const Tensor& pose, double eps, bool with_plus,
                        int64_t stream, const OptTensor& delta_stats, double delta_rel) {
    (void)x3d; (void)x2d; (void)w2d;
    ProblemTensors pt;
    pt.x3d = x3d_c; pt.x2d = x2d_c; pt.w2d = w2d_c; pt.cam = cam_c; pt.delta = delta_c;
    if (lb_c.has_value() && ub_c.has_value()) { pt.lb = *lb_c; pt.ub = *ub_c; }
    if (status.has_value()) pt.status = *status;
    pt.z_min = z_min; pt.huber_eps = huber_eps; pt.dof = dof;
    const epropnp_problem prob = pt.c();
    const Tensor ps = pose.detach().contiguous();
    const int64_t B = prob.num_obj;
    Tensor out = torch::empty({B, with_plus ? (dof == 6 ? 7 : 4) : dof}, x2d_c.options());
    check(with_plus ? epropnp_pose_opt_plus_forward(&prob, (float)eps, fptr(ps), fptr(out), (void*)stream)
                    : epropnp_gn_step_forward(&prob, (float)eps, fptr(ps), fptr(out), (void*)stream), "epropnp_gn_step_forward");
    ctx->save_for_backward({ps});
    ctx->saved_data["x3d"] = x3d_c; ctx->saved_data["x2d"] = x2d_c; ctx->saved_data["w2d"] = w2d_c;
    ctx->saved_data["cam"] = cam_c; ctx->saved_data["delta"] = delta_c;
    ctx->saved_data["lb"] = pt.lb.defined() ? c10::IValue(pt.lb) : c10::IValue();
    ctx->saved_data["ub"] = pt.ub.defined() ? c10::IValue(pt.ub) : c10::IValue();
    ctx->saved_data["z_min"] = z_min; ctx->saved_data["huber_eps"] = huber_eps; ctx->saved_data["dof"] = dof;
    ctx->saved_data["eps"] = eps; ctx->saved_data["with_plus"] = with_plus; ctx->saved_data["stream"] = stream;
    const bool fold = delta_stats.has_value() && delta_stats->defined();     // epropnp_problem.delta_stats: see FusedMonteCarlo
    ctx->saved_data["dstats"] = fold ? c10::IValue(delta_stats->detach().contiguous()) : c10::IValue();
    ctx->saved_data["drel"] = delta_rel;
    ctx->saved_data["delta_dim"] = (delta.has_value() && delta->defined()) ? (int64_t)delta->dim() : (int64_t)-1;
    std::vector<int64_t> vers;
    for (const Tensor* t : {&x3d_c, &x2d_c, &w2d_c, &delta_c, &cam_c}) vers.push_back((int64_t)t->_version());
    ctx->saved_data["versions"] = vers;
    ctx->set_materialize_grads(false);
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list out(21);
    if (!grads[0].defined()) return out;
    ProblemTensors pt;
    pt.x3d = ctx->saved_data["x3d"].toTensor(); pt.x2d = ctx->saved_data["x2d"].toTensor();
    pt.w2d = ctx->saved_data["w2d"].toTensor(); pt.cam = ctx->saved_data["cam"].toTensor();
    pt.delta = ctx->saved_data["delta"].toTensor();
    if (!ctx->saved_data["lb"].isNone()) { pt.lb = ctx->saved_data["lb"].toTensor(); pt.ub = ctx->saved_data["ub"].toTensor(); }
    pt.z_min = ctx->saved_data["z_min"].toDouble(); pt.huber_eps = ctx->saved_data["huber_eps"].toDouble();
    const bool fold = !ctx->saved_data["dstats"].isNone();
    if (fold) { pt.dstats = ctx->saved_data["dstats"].toTensor(); pt.drel = ctx->saved_data["drel"].toDouble(); }
    pt.dof = ctx->saved_data["dof"].toInt();
    const auto vers = ctx->saved_data["versions"].toIntVector();
    const Tensor* ts[5] = {&pt.x3d, &pt.x2d, &pt.w2d, &pt.delta, &pt.cam};
    for (int i = 0; i < 5; ++i)
      TORCH_CHECK((int64_t)ts[i]->_version() == vers[i],
                  "one of the variables needed for gradient computation has been modified by an inplace operation "
                  "(EPro-PnP recomputes its backward from x3d / x2d / w2d / delta / cam_mats: they must stay unchanged "
                  "between forward and backward)");
    const epropnp_problem q = pt.c();
    const Tensor ps = ctx->get_saved_variables()[0], g = grads[0].contiguous();
    const int64_t B = q.num_obj, N = q.num_pts;
    const auto opt = ps.options();
    Tensor gx3d = torch::empty({B, N, 3}, opt), gx2d = torch::empty({B, N, 2}, opt), gw2d = torch::empty({B, N, 2}, opt),
           gdel = torch::empty({B}, opt);
    const float eps = (float)ctx->saved_data["eps"].toDouble();
    void* st = (void*)ctx->saved_data["stream"].toInt();
    check(ctx->saved_data["with_plus"].toBool()
              ? epropnp_pose_opt_plus_backward(&q, eps, fptr(ps), fptr(g), fptr(gx3d), fptr(gx2d), fptr(gw2d), fptr(gdel), st)
              : epropnp_gn_step_backward(&q, eps, fptr(ps), fptr(g), fptr(gx3d), fptr(gx2d), fptr(gw2d), fptr(gdel), st),
          "epropnp_gn_step_backward");
    if (ctx->needs_input_grad(0)) out[0] = gx3d;
    if (ctx->needs_input_grad(1)) out[1] = gx2d;
    if (ctx->needs_input_grad(2)) out[2] = gw2d;
    const int64_t ddim = ctx->saved_data["delta_dim"].toInt();
    if (!fold && ddim >= 0 && ctx->needs_input_grad(3)) out[3] = (ddim == 0) ? gdel.sum() : gdel;
    return out;
  }
};

// pose translation += sign * R(pose) offset (pnp_normalize / pnp_denormalize, common.py:118-136); differentiable w.r.t. pose
struct ShiftPoses : public torch::autograd::Function<ShiftPoses> {
  static Tensor forward(AutogradContext* ctx, const Tensor& pose, const Tensor& offset, double sign, int64_t stream) {
    const Tensor ps = pose.detach().contiguous(), off = offset.detach().contiguous();
    const int64_t B = ps.size(-2), PL = ps.size(-1), P = ps.numel() / (B * PL);
    Tensor out = torch::empty_like(ps);
    check(epropnp_shift_poses(fptr(ps), fptr(off), (int32_t)P, (int32_t)B, PL == 7 ? 6 : 4, (float)sign, fptr(out),
                              (void*)stream), "epropnp_shift_poses");
    ctx->save_for_backward({ps, off});
    ctx->saved_data["sign"] = sign; ctx->saved_data["stream"] = stream;
    ctx->set_materialize_grads(false);
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    if (!grads[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor()};
    const auto saved = ctx->get_saved_variables();
    const Tensor &ps = saved[0], &off = saved[1];
    const Tensor g = grads[0].contiguous();
    const int64_t B = ps.size(-2), PL = ps.size(-1), P = ps.numel() / (B * PL);
    Tensor gp = torch::empty_like(ps);
    check(epropnp_shift_poses_backward(fptr(ps), fptr(off), fptr(g), (int32_t)P, (int32_t)B, PL == 7 ? 6 : 4,
                                       (float)ctx->saved_data["sign"].toDouble(), fptr(gp),
                                       (void*)ctx->saved_data["stream"].toInt()), "epropnp_shift_poses_backward");
    return {gp, Tensor(), Tensor(), Tensor()};
  }
};

// ---- Python-facing wrappers ------------------------------------------------------------------------------------------
Tensor gn_step(const Tensor& x3d, const Tensor& x2d, const Tensor& w2d, const OptTensor& delta, const Tensor& x3d_c,
               const Tensor& x2d_c, const Tensor& w2d_c, const Tensor& cam_c, const OptTensor& lb_c, const OptTensor& ub_c,
               const Tensor& delta_c, const OptTensor& status, double z_min, double huber_eps, int64_t dof, const Tensor& pose,
               double eps, bool with_plus, int64_t stream, const OptTensor& delta_stats, double delta_rel) {
  return GnStep::apply(x3d, x2d, w2d, delta, x3d_c, x2d_c, w2d_c, cam_c, lb_c, ub_c, delta_c, status, z_min, huber_eps, dof, pose,
                       eps, with_plus, stream, delta_stats, delta_rel);
}

Tensor shift_poses(const Tensor& pose, const Tensor& offset, double sign, int64_t stream) {
  return ShiftPoses::apply(pose, offset, sign, stream);
}

variable_list adaptive_delta(const Tensor& x2d, const Tensor& w2d, double rel, int64_t stream) {
  return AdaptiveDelta::apply(x2d, w2d, rel, stream);       // (delta, stats)
}

Tensor mc_pose_loss(const Tensor& logw, const OptTensor& cost_target, int64_t stream) {
  return McPoseLoss::apply(logw, cost_target, stream);
}

Tensor mc_pose_loss_reduced(const Tensor& logw, const OptTensor& cost_target, const OptTensor& weight, double scale,
                            double momentum, const OptTensor& nf_in, const OptTensor& norm_factor, int64_t stream) {
  return McPoseLossReduced::apply(logw, cost_target, weight, scale, momentum, nf_in, norm_factor, stream);
}

std::vector<OptTensor> fused_monte_carlo(const Tensor& x3d, const Tensor& x2d, const Tensor& w2d, const OptTensor& delta,
                                         const Tensor& x3d_c, const Tensor& x2d_c, const Tensor& w2d_c, const Tensor& cam_c,
                                         const OptTensor& lb_c, const OptTensor& ub_c, const Tensor& delta_c,
                                         const OptTensor& status, double z_min, double huber_eps, int64_t dof,
                                         const OptTensor& pose_init, const OptTensor& noise, const py::bytes& mc_params,
                                         bool with_cost, int64_t nsplit, int64_t stream, const OptTensor& delta_stats,
                                         double delta_rel) {
  const variable_list r = FusedMonteCarlo::apply(x3d, x2d, w2d, delta, x3d_c, x2d_c, w2d_c, cam_c, lb_c, ub_c, delta_c, status,
                                                 z_min, huber_eps, dof, pose_init, noise, std::string(mc_params), with_cost,
                                                 nsplit, stream, delta_stats, delta_rel);
  std::vector<OptTensor> out;
  for (const Tensor& t : r) out.push_back((t.defined() && t.numel() > 0) ? OptTensor(t) : OptTensor());
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd nodes of the EPro-PnP layer over libepropnp_hip.so (include/epropnp_hip.h)";
  m.def("abi_version", []() { return epropnp_abi_version(); });
  m.def("mc_params_size", []() { return (int64_t)sizeof(epropnp_mc_params); });
  m.def("adaptive_delta", &adaptive_delta);
  m.def("mc_pose_loss", &mc_pose_loss);
  m.def("mc_pose_loss_reduced", &mc_pose_loss_reduced);
  m.def("fused_monte_carlo", &fused_monte_carlo);
  m.def("gn_step", &gn_step);
  m.def("shift_poses", &shift_poses);
}
