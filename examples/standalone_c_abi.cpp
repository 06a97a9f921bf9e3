// standalone_c_abi.cpp -- the library without PyTorch: a plain HIP host program that allocates device buffers, builds a
// small synthetic batch (the generator of SURVEY.md section 8d), and runs the hot path through the C ABI only:
//   epropnp_adaptive_delta -> epropnp_lm_solve -> epropnp_amis_forward -> epropnp_mc_loss_forward/backward ->
//   epropnp_amis_backward.
// Build:  g++ -O2 -std=c++17 -I include -I /opt/rocm/include examples/standalone_c_abi.cpp -o /tmp/standalone
//             -L epro-pnp_amd/lib -lepropnp_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/epro-pnp_amd/lib -Wl,-rpath,/opt/rocm/lib
// (tests/test_c_abi.py::test_standalone_program_on_gpu does exactly this on the GPU box.)
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "epropnp_hip.h"

#define HIP_OK(x)                                                                 \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)
#define PNP_OK(x)                                                                                    \
  do {                                                                                               \
    if ((x) != EPROPNP_OK) { std::fprintf(stderr, "%s: %s\n", #x, epropnp_last_error()); return 3; } \
  } while (0)

template <class T>
static T* to_device(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}
template <class T>
static T* device_alloc(size_t n) {
  T* d = nullptr;
  return hipMalloc(&d, n * sizeof(T)) == hipSuccess ? d : nullptr;
}

int main() {
  const int B = 64, N = 128, S = 128, K = 4, L = 3;
  std::mt19937 rng(0);
  std::normal_distribution<float> nrm(0.f, 1.f);
  std::uniform_real_distribution<float> uni(0.f, 1.f);
  std::vector<float> x3d(B * N * 3), x2d(B * N * 2), w2d(B * N * 2), cam(B * 9), pose_gt(B * 7), pose_init(B * 7);
  const float Kmat[9] = {800, 0, 320, 0, 800, 240, 0, 0, 1};
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < 9; ++i) cam[b * 9 + i] = Kmat[i];
    float t[3] = {nrm(rng), nrm(rng), nrm(rng) + 5.f}, q[4], qn = 0.f;
    for (float& v : q) { v = nrm(rng); qn += v * v; }
    for (float& v : q) v /= std::sqrt(qn);
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    float wsum[2] = {0.f, 0.f};
    for (int n = 0; n < N; ++n) {
      float X[3], c[3];
      for (int i = 0; i < 3; ++i) X[i] = x3d[(b * N + n) * 3 + i] = 0.5f * nrm(rng);
      for (int i = 0; i < 3; ++i) c[i] = R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2] + t[i];
      x2d[(b * N + n) * 2] = 800.f * c[0] / c[2] + 320.f + nrm(rng);
      x2d[(b * N + n) * 2 + 1] = 800.f * c[1] / c[2] + 240.f + nrm(rng);
      for (int i = 0; i < 2; ++i) wsum[i] += (w2d[(b * N + n) * 2 + i] = std::exp(uni(rng)));
    }
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < 2; ++i) w2d[(b * N + n) * 2 + i] *= 2.f / wsum[i];          // softmax_N(U(0,1)) * 2
    float qi[4], qin = 0.f;
    for (int i = 0; i < 3; ++i) { pose_gt[b * 7 + i] = t[i]; pose_init[b * 7 + i] = t[i] + 0.1f * nrm(rng); }
    for (int i = 0; i < 4; ++i) { pose_gt[b * 7 + 3 + i] = q[i]; qi[i] = q[i] + 0.05f * nrm(rng); qin += qi[i] * qi[i]; }
    for (int i = 0; i < 4; ++i) pose_init[b * 7 + 3 + i] = qi[i] / std::sqrt(qin);
  }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::fprintf(stderr, "no HIP device\n"); return 1; }
  float *d_x3d = to_device(x3d), *d_x2d = to_device(x2d), *d_w2d = to_device(w2d), *d_cam = to_device(cam);
  float *d_init = to_device(pose_init), *d_gt = to_device(pose_gt), *d_cgt = device_alloc<float>(B), *d_delta = device_alloc<float>(B), *d_stats = device_alloc<float>(B * 4);
  float *d_opt = device_alloc<float>(B * 7), *d_cov = device_alloc<float>(B * 36), *d_cost = device_alloc<float>(B);
  float *d_cinit = device_alloc<float>(B), *d_smp = device_alloc<float>((size_t)S * B * 7), *d_logw = device_alloc<float>((size_t)S * B);
  float *d_loss = device_alloc<float>(B), *d_lse = device_alloc<float>(B), *d_gl = to_device(std::vector<float>(B, 1.0f / B));
  float *d_glogw = device_alloc<float>((size_t)S * B), *d_gct = device_alloc<float>(B);
  float *d_gx3d = device_alloc<float>(B * N * 3), *d_gx2d = device_alloc<float>(B * N * 2), *d_gw2d = device_alloc<float>(B * N * 2);
  float* d_gdel = device_alloc<float>(B);
  if (!d_x3d || !d_gdel) { std::fprintf(stderr, "device allocation failed\n"); return 2; }

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  PNP_OK(epropnp_adaptive_delta(d_x2d, d_w2d, B, N, 0.5f, d_delta, d_stats, st));
  epropnp_problem prob = {d_x3d, d_x2d, d_w2d, d_cam, nullptr, nullptr, d_delta, 0.1f, B, N, 6};
  epropnp_lm_params lm = {L, 0, 1e-6f, 1e32f, 1e-3f, 30.0f, 1e16f, 1e-5f};
  PNP_OK(epropnp_evaluate_cost(&prob, d_init, 1, d_cinit, st));
  PNP_OK(epropnp_evaluate_cost(&prob, d_gt, 1, d_cgt, st));       // cost_target of the Monte-Carlo pose loss
  PNP_OK(epropnp_lm_solve(&prob, &lm, d_init, d_opt, d_cov, d_cost, nullptr, nullptr, 0, st));
  epropnp_amis_params amis = {S, K, 1e-5f, 3, 0.001f, 1234u, 0u, nullptr};
  PNP_OK(epropnp_amis_forward(&prob, &amis, d_opt, d_cov, nullptr, d_smp, d_logw, nullptr, st));
  PNP_OK(epropnp_mc_loss_forward(d_logw, d_cgt, S, B, d_loss, d_lse, st));
  PNP_OK(epropnp_mc_loss_backward(d_logw, d_lse, d_loss, d_gl, S, B, d_glogw, d_gct, st));
  PNP_OK(epropnp_amis_backward(&prob, d_smp, d_glogw, S, d_gt, d_gct, d_gx3d, d_gx2d, d_gw2d, d_gdel, st));
  HIP_OK(hipStreamSynchronize(st));

  std::vector<float> opt(B * 7), loss(B), gx(B * N * 3), cost(B), cinit(B);
  HIP_OK(hipMemcpy(opt.data(), d_opt, opt.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(loss.data(), d_loss, loss.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(gx.data(), d_gx3d, gx.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(cost.data(), d_cost, cost.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(cinit.data(), d_cinit, cinit.size() * 4, hipMemcpyDeviceToHost));
  double terr = 0, mloss = 0, gnorm = 0;
  int improved = 0, finite = 1;
  for (int b = 0; b < B; ++b) {
    double e = 0;
    for (int i = 0; i < 3; ++i) e += std::pow(opt[b * 7 + i] - pose_gt[b * 7 + i], 2);
    terr += std::sqrt(e) / B;
    mloss += loss[b] / B;
    improved += cost[b] <= cinit[b] * (1 + 1e-5f) + 1e-6f;
    finite &= std::isfinite(loss[b]) ? 1 : 0;
  }
  for (float v : gx) { gnorm += (double)v * v; finite &= std::isfinite(v) ? 1 : 0; }
  std::printf("STANDALONE objects=%d mean_translation_error=%.5f mean_mc_loss=%.5f lm_not_worse=%d/%d |grad_x3d|=%.6f finite=%d\n",
              B, terr, mloss, improved, B, std::sqrt(gnorm), finite);
  return (finite && improved == B && terr < 0.05) ? 0 : 4;
}
