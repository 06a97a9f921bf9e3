// Micro-benchmark: do fp32 MFMAs (v_mfma_f32_16x16x4_f32, the projection of the AMIS sweeps) overlap with fp32 VALU work on
// the same SIMD?  One loop iteration = one 16x16 (pose, point) tile of the forward sweep in instruction mix:
//   3 independent MFMAs  +  32 v_pk_fma_f32 (or 64 v_fma_f32) on 8 independent register chains.
// Modes: VALU only, MFMA only, both in one wave (MFMAs first, as the compiler schedules them).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define X4(S) S S S S
#define X8(S) S S S S S S S S

template <bool VALU, int MFMA, bool PACKED>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  const f2 A = {a, a}, Bv = {b, b};
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
  float ma = a * threadIdx.x, mb = b;
  const f4 wa = {ma, mb, ma, mb}, wb = {mb, ma, mb, ma};
  const f2 ha = {ma, mb}, hb = {mb, ma};
  for (int i = 0; i < iters; ++i) {
    if (MFMA == 2)      // gfx950 bf16 MFMA, K = 32: A and B are 8 bf16 (4 VGPRs) per lane
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n v_mfma_f32_16x16x32_bf16 %1, %3, %4, %1\n v_mfma_f32_16x16x32_bf16 %2, %3, %4, %2\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(wa), "v"(wb));
    if (MFMA == 3)      // bf16 MFMA, K = 16: 4 bf16 (2 VGPRs) per lane
      asm volatile("v_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n v_mfma_f32_16x16x16_bf16 %1, %3, %4, %1\n v_mfma_f32_16x16x16_bf16 %2, %3, %4, %2\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(ha), "v"(hb));
    if (MFMA == 1)
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %3, %4, %0\n v_mfma_f32_16x16x4_f32 %1, %3, %4, %1\n v_mfma_f32_16x16x4_f32 %2, %3, %4, %2\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(ma), "v"(mb));
    if (VALU && PACKED)
      asm volatile(X4("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                      "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(A), "v"(Bv));
    if (VALU && !PACKED)
      asm volatile(X8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x +
                                               p7.y + c0.x + c1.y + c2.z;
}

template <bool VALU, int MFMA, bool PACKED>
double run(int waves_per_simd) {
  float* out;
  const int blocks = 256 * waves_per_simd, threads = 256, iters = 4000;
  hipMalloc(&out, blocks * threads * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<VALU, MFMA, PACKED><<<blocks, threads>>>(out, 50, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<VALU, MFMA, PACKED><<<blocks, threads>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms * 1e6 / ((double)waves_per_simd * iters);   // ns per tile-equivalent per SIMD
}

int main() {
  printf("ns per (3 MFMA + VALU block) per SIMD; 'sum' = no overlap, 'max' = perfect overlap\n");
  const char* names[4] = {"", "f32 16x16x4", "bf16 16x16x32", "bf16 16x16x16"};
  for (int w : {1, 2, 4}) {
    const double v = run<true, 0, true>(w), vs = run<true, 0, false>(w);
    const double m[4] = {0, run<false, 1, true>(w), run<false, 2, true>(w), run<false, 3, true>(w)};
    const double both[4] = {0, run<true, 1, true>(w), run<true, 2, true>(w), run<true, 3, true>(w)};
    const double boths[4] = {0, run<true, 1, false>(w), run<true, 2, false>(w), run<true, 3, false>(w)};
    for (int i = 1; i < 4; ++i) {
      printf("w/SIMD=%d  %-14s + 32 v_pk_fma: valu %6.1f  mfma %6.1f  both %6.1f  (sum %6.1f, max %6.1f)\n", w, names[i], v, m[i], both[i],
             v + m[i], v > m[i] ? v : m[i]);
      printf("w/SIMD=%d  %-14s + 64 v_fma   : valu %6.1f  mfma %6.1f  both %6.1f  (sum %6.1f, max %6.1f)\n", w, names[i], vs, m[i], boths[i],
             vs + m[i], vs > m[i] ? vs : m[i]);
    }
  }
  return 0;
}
