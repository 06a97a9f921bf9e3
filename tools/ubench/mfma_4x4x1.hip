// Micro-benchmark for v_mfma_f32_4x4x1_16b_f32 (16 blocks of a 4 x 4 outer product, K = 1), the candidate for the
// back-projection  g_x3d[n] += sum_j (K R)_j^T g_h[j,n]  of the AMIS backward:
//   1. operand / result layout:  D[i][l] += A[4 (l >> 2) + i] * B[l]   (vgpr i, lane l)  -- checked against a host loop
//   2. issue cost per instruction per SIMD: 12 of them accumulating into ONE 4-register D (the back-projection of one
//      16 x 16 tile) vs into 3 independent Ds, vs the 36 v_fma_f32 they would replace, at 1 / 2 / 3 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_4x4x1.hip -o /tmp/mfma_4x4x1 && /tmp/mfma_4x4x1
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[i * 64 + l] = c[i];
}

#define X4(S) S S S S
#define X12(S) S S S S S S S S S S S S
template <int MODE>
__global__ void rate(float* out, int iters, float a, float b) {
  float va = a * threadIdx.x, vb = b + threadIdx.x;
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0)        // 12 MFMAs, one accumulator (dependent chain)
      asm volatile(X12("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n") : "+v"(c0) : "v"(va), "v"(vb));
    if (MODE == 1)        // 12 MFMAs over three accumulators
      asm volatile(X4("v_mfma_f32_4x4x1_16b_f32 %0, %3, %4, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %3, %4, %1\n v_mfma_f32_4x4x1_16b_f32 %2, %3, %4, %2\n")
                   : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(va), "v"(vb));
    if (MODE == 2)        // the VALU work they replace: 36 v_fma_f32 on three accumulators (9 per point-pose x 4)
      asm volatile(X12("v_fma_f32 %0, %3, %4, %0\n v_fma_f32 %1, %3, %4, %1\n v_fma_f32 %2, %3, %4, %2\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(va), "v"(vb));
    if (MODE == 3)        // 3 x 16x16x4 (the projection of one tile), for scale
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %3, %4, %0\n v_mfma_f32_16x16x4_f32 %1, %3, %4, %1\n v_mfma_f32_16x16x4_f32 %2, %3, %4, %2\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(va), "v"(vb));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0.x + c1.y + c2.z + x0 + x1 + x2;
}

template <int MODE>
double run(int waves_per_simd) {
  float* out;
  const int blocks = 256 * waves_per_simd, threads = 256, iters = 4000;
  hipMalloc(&out, blocks * threads * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate<MODE><<<blocks, threads>>>(out, 50, 1.0001f, 0.5f);
  hipEventRecord(e0);
  rate<MODE><<<blocks, threads>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms * 1e6 / ((double)waves_per_simd * iters);   // ns per loop body per SIMD
}

int main() {
  float ha[64], hb[64], hd[256], *a, *b, *d;
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  layout<<<1, 64>>>(a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 4; ++i)
    for (int l = 0; l < 64; ++l) bad += hd[i * 64 + l] != ha[4 * (l >> 2) + i] * hb[l];
  printf("layout D[i][l] = A[4 (l >> 2) + i] * B[l]: %s (%d mismatches)\n", bad ? "NO" : "confirmed", bad);
  if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, hd[l], hd[64 + l], hd[128 + l], hd[192 + l]);
  printf("ns per loop body per SIMD (12 x 4x4x1 one acc | 12 x 4x4x1 three acc | 36 v_fma_f32 | 3 x 16x16x4)\n");
  for (int w : {1, 2, 3, 4})
    printf("w/SIMD=%d   %7.1f   %7.1f   %7.1f   %7.1f\n", w, run<0>(w), run<1>(w), run<2>(w), run<3>(w));
  return 0;
}
