// tuning.h -- the ONE place where tuning builds differ from the product (build.py -D PNP_TUNING [-D PNP_TUNING_REFIT] --tag <t>,
// driven by tools/tune.py).  Production builds see empty statements and a compile-time `false`; the kernel files carry no
// conditional compilation of their own.  What the A/B runs of earlier rounds decided stays on file under profiles/ (HISTORY.md);
// the losing code paths are gone.
#pragma once

// ---- compile-time constants a tuning variant may override (-D NAME=value) ---------------------------------------------------
#ifndef PNP_FIT_T          // precision of the single-lane proposal fits (float measures what fp64 costs there: 2.3 % of the forward)
#define PNP_FIT_T double
#endif
#ifndef PNP_FIT_FN         // the fp64 fits are inlined by default (an out-of-line variant was measured: no gain)
#define PNP_FIT_FN __device__ __forceinline__
#endif
#ifndef PNP_VM_TRIES       // attempts of the bounded Best-Fisher von Mises sampler
#define PNP_VM_TRIES 16
#endif
#ifndef PNP_FWD_BF16_MINW  // waves per SIMD the split-projection 6-DoF forward with <= 8 resident tiles is compiled for
#define PNP_FWD_BF16_MINW 3
#endif
#ifndef PNP_FWD_PIPE       // 1: the forward's sweep issues tile i + 1's projections in front of tile i's Huber sweep (round 6: -4.7 % at C2)
#define PNP_FWD_PIPE 1
#endif
#ifndef PNP_FWD_MINW       // waves per SIMD the fp32-projection 6-DoF forward is compiled for (the bf16 one: 3)
#define PNP_FWD_MINW 4
#endif
#ifndef PNP_BWD_MINW4      // ... with four resident point tiles (few objects, or pose tables that leave room for two workgroups per CU only)
#define PNP_BWD_MINW4 2
#endif
#ifndef PNP_BWD_MINW       // waves per SIMD of the MFMA backward with <= 2 resident point tiles
#define PNP_BWD_MINW 3
#endif

// ---- instrumentation ----------------------------------------------------------------------------------------------------------
#ifdef PNP_TUNING
// phase ablation: `par.ablate` bits switch phases of the forward off (bit0 sweep, bit1 proposal refit, bit2 densities, bit3 moment
// pass, bit4 its reductions, bit5 re-draw after the first iteration)
#define PNP_ABLATED(par, bit) ((((par).ablate) & (bit)) != 0)
// shader-clock cycles per phase of a kernel, accumulated by thread 0 in registers and flushed once per workgroup (per-phase atomics
// on one address doubled the kernel time)
#define PNP_PHASES_BEGIN(n)                \
  long long phase_t0_ = clock64();         \
  unsigned long long phase_acc_[n] = {}
#define PNP_PHASE(i)                                                   \
  do {                                                                 \
    if (tid == 0) {                                                    \
      const long long now_ = clock64();                                \
      phase_acc_[i] += (unsigned long long)(now_ - phase_t0_);         \
      phase_t0_ = now_;                                                \
    }                                                                  \
  } while (0)
#define PNP_PHASES_FLUSH(n)                                                                      \
  do {                                                                                           \
    if (tid == 0) {                                                                              \
      for (int i_ = 0; i_ < (n); ++i_) atomicAdd(&::pnp::tuning::g_phase[i_], phase_acc_[i_]);   \
    }                                                                                            \
  } while (0)
#else
#define PNP_ABLATED(par, bit) false
#define PNP_PHASES_BEGIN(n)
#define PNP_PHASE(i)
#define PNP_PHASES_FLUSH(n)
#endif

#if defined(PNP_TUNING) && defined(PNP_TUNING_REFIT)
// cycles of the fitting lane inside the proposal refit [moment pass + reductions | ACG fixed-point iterations | final fits | draws]
// (per-phase atomics: they perturb the kernel, hence their own switch)
#define PNP_REFIT_BEGIN() long long refit_t0_ = clock64()
#define PNP_REFIT_PHASE(i)                                                                      \
  do {                                                                                          \
    if (tid == 0) {                                                                             \
      const long long now_ = clock64();                                                         \
      atomicAdd(&::pnp::tuning::g_refit_phase[i], (unsigned long long)(now_ - refit_t0_));      \
      refit_t0_ = now_;                                                                         \
    }                                                                                           \
  } while (0)
#define PNP_REFIT_CLOCK(name) const long long name = clock64()
#define PNP_REFIT_ADD(i, name)                                                                                      \
  do {                                                                                                              \
    if (tid == 0) atomicAdd(&::pnp::tuning::g_refit_phase[i], (unsigned long long)(clock64() - (name)));            \
  } while (0)
#else
#define PNP_REFIT_BEGIN()
#define PNP_REFIT_PHASE(i)
#define PNP_REFIT_CLOCK(name)
#define PNP_REFIT_ADD(i, name)
#endif

namespace pnp {
namespace tuning {
#ifdef PNP_TUNING
// Per translation unit (the library is built without relocatable device code): the kernel of a .hip file fills ITS copy and the
// reader below, instantiated in the same file, reads it back.
static __device__ unsigned long long g_phase[8];
static __device__ unsigned long long g_refit_phase[4];
// out[0..5]: the kernel's phases; with_refit: out[6] = ACG fixed-point iterations, out[7] = final fits of the proposal refit
static inline int read_cycles(unsigned long long* out, int reset, bool with_refit) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (with_refit && out) {
    unsigned long long rf[4];
    if (hipMemcpyFromSymbol(rf, HIP_SYMBOL(g_refit_phase), sizeof(rf)) != hipSuccess) return -1;
    out[6] = rf[1];
    out[7] = rf[2];
  }
  if (reset) {
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_refit_phase), z, sizeof(unsigned long long) * 4) != hipSuccess) return -1;
  }
  return 0;
}
#else
static inline int read_cycles(unsigned long long*, int, bool) { return -1; }      // not a tuning build
#endif
}  // namespace tuning
}  // namespace pnp
