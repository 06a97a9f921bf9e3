// Are packed fp32 instructions with op_sel modifiers reliable on gfx950 while the SIMD's matrix pipe is busy?
//
// Round 5: builds of amis_backward_mfma_kernel returned wrong, run-to-run different gradients whenever two or more waves shared a
// SIMD (profiles/r05_bwd_scratch.txt).  Replacing -- in the ASSEMBLY of a failing build, nothing else touched -- the 91 v_pk_{mul,add,
// fma}_f32 instructions that carry op_sel / op_sel_hi modifiers by pairs of scalar fp32 instructions makes that build bit-identical to
// the passing one; replacing the 143 packed instructions WITHOUT modifiers does not.  Other kernels of the library carry the same
// modifiers and are clean, so something else has to coincide.  This probe looks for it in isolation: victim waves execute
//     v_pk_mul_f32 D, A, B op_sel_hi:[1,0]      (D.lo = A.lo B.lo, D.hi = A.hi B.lo: the broadcast form the SLP vectoriser emits)
//     v_pk_fma_f32 D, A, B, C op_sel:[1,0,0]    (both halves read A.hi)
// on lane-dependent data, K instructions behind an MFMA of their own (or none), and compare with the same products from scalar
// instructions; the other half of the waves (waves 4-7, 12-15 of a workgroup) issue nothing but MFMAs.
// hipcc --offload-arch=gfx950 -O2 -w tools/ubench/pk_opsel_under_mfma.hip -o tools/ubench/pk_opsel_under_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define BG4 "v_mfma_f32_16x16x32_bf16 v[60:63], %1, %2, 0\n\tv_mfma_f32_16x16x32_bf16 v[64:67], %2, %1, 0\n\t" \
            "v_mfma_f32_16x16x32_bf16 v[68:71], %1, %2, 0\n\tv_mfma_f32_16x16x32_bf16 v[72:75], %2, %1, 0\n\t"

// OWN: 0 = the victim issues no MFMA of its own, 1 = one MFMA directly in front of the packed instructions, 2 = four of them
template <int OWN>
__global__ void kern(const u32x4* __restrict__ A, const u32x4* __restrict__ B, const float* __restrict__ X, unsigned* __restrict__ bad, int iters,
                     int hammer) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const u32x4 a = A[gid & 4095], b = B[(gid * 3) & 4095];
  if (hammer && ((threadIdx.x >> 8) & 1)) {
    float acc = 0.f;
    if (hammer == 1) {
      for (int it = 0; it < 3 * iters; ++it)
        asm volatile(BG4 BG4 BG4 BG4 BG4 BG4 BG4 BG4 : "=v"(acc) : "v"(a), "v"(b)
                     : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");
    } else {      // transcendental unit: back-to-back v_rsq / v_rcp on private registers
      float t = X[gid & 65535] + 3.0f;
      for (int it = 0; it < 12 * iters; ++it)
        asm volatile("v_rsq_f32 v60, %0\n\tv_rcp_f32 v61, %0\n\tv_rsq_f32 v62, %0\n\tv_rcp_f32 v63, %0\n\tv_rsq_f32 v64, %0\n\tv_rcp_f32 v65, %0\n\t"
                     "v_rsq_f32 v66, %0\n\tv_rcp_f32 v67, %0\n\t" : "+v"(t) :: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
      if (t == 12345.f) bad[gid] = 1;
    }
    return;
  }
  unsigned nbad = 0;
  float x0 = X[gid & 65535], x1 = X[(gid + 7) & 65535], y0 = X[(gid * 5 + 1) & 65535], y1 = X[(gid * 11 + 3) & 65535];
  float c0 = X[(gid * 13 + 5) & 65535], c1 = X[(gid * 17 + 9) & 65535];
  for (int it = 0; it < iters; ++it) {
    float m0, m1, f0, f1, n0, n1;
    // v[80:81] = A pair (x0, x1), v[82:83] = B pair (y0, y1), v[84:85] = C pair
    asm volatile("v_mov_b32 v80, %6\n\tv_mov_b32 v81, %7\n\tv_mov_b32 v82, %8\n\tv_mov_b32 v83, %9\n\tv_mov_b32 v84, %10\n\tv_mov_b32 v85, %11\n\t"
                 "s_nop 7\n\t"
                 ".if %14 == 1\n\t v_mfma_f32_16x16x32_bf16 v[60:63], %12, %13, 0\n\t.endif\n\t"
                 ".if %14 == 2\n\t v_mfma_f32_16x16x32_bf16 v[60:63], %12, %13, 0\n\tv_mfma_f32_16x16x32_bf16 v[64:67], %13, %12, 0\n\t"
                 "v_mfma_f32_16x16x32_bf16 v[68:71], %12, %13, 0\n\tv_mfma_f32_16x16x32_bf16 v[72:75], %13, %12, 0\n\t.endif\n\t"
                 ".if %14 == 3\n\t v_rsq_f32 v60, %6\n\tv_rsq_f32 v61, %7\n\tv_rcp_f32 v62, %8\n\tv_rcp_f32 v63, %9\n\t.endif\n\t"
                 "v_pk_mul_f32 v[86:87], v[80:81], v[82:83] op_sel_hi:[1,0]\n\t"
                 "v_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0]\n\t"
                 "v_pk_mul_f32 v[90:91], v[80:81], v[82:83] op_sel:[0,1]\n\t"
                 "s_nop 7\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v86\n\tv_mov_b32 %1, v87\n\tv_mov_b32 %2, v88\n\tv_mov_b32 %3, v89\n\tv_mov_b32 %4, v90\n\tv_mov_b32 %5, v91\n\t"
                 : "=&v"(m0), "=&v"(m1), "=&v"(f0), "=&v"(f1), "=&v"(n0), "=&v"(n1)
                 : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1), "v"(a), "v"(b), "n"(OWN)
                 : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v80", "v81",
                   "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93");
    // expected: scalar instructions
    float e0, e1, g0, g1, h0, h1;
    asm volatile("v_mul_f32 %0, %6, %8\n\tv_mul_f32 %1, %7, %8\n\tv_fma_f32 %2, %7, %8, %10\n\tv_fma_f32 %3, %7, %9, %11\n\t"
                 "v_mul_f32 %4, %6, %9\n\tv_mul_f32 %5, %7, %9\n\ts_nop 1\n\t"
                 : "=&v"(e0), "=&v"(e1), "=&v"(g0), "=&v"(g1), "=&v"(h0), "=&v"(h1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    nbad += (__float_as_uint(m0) != __float_as_uint(e0)) + (__float_as_uint(m1) != __float_as_uint(e1)) +
            (__float_as_uint(f0) != __float_as_uint(g0)) + (__float_as_uint(f1) != __float_as_uint(g1)) +
            (__float_as_uint(n0) != __float_as_uint(h0)) + (__float_as_uint(n1) != __float_as_uint(h1));
    x0 = x0 * 1.0000001f + 0.25f; y1 = y1 * 0.9999999f - 0.125f;      // fresh operands every trip
  }
  bad[gid] = nbad;
}

template <int OWN>
static unsigned long long run(const u32x4* dA, const u32x4* dB, const float* dX, unsigned* dBad, int blocks, int threads, int hammer) {
  hipMemset(dBad, 0, (size_t)blocks * threads * 4);
  hipLaunchKernelGGL((kern<OWN>), dim3(blocks), dim3(threads), 0, 0, dA, dB, dX, dBad, 2000, hammer);
  std::vector<unsigned> h((size_t)blocks * threads);
  hipMemcpy(h.data(), dBad, h.size() * 4, hipMemcpyDeviceToHost);
  unsigned long long s = 0;
  for (unsigned v : h) s += v;
  return s;
}

int main() {
  std::vector<unsigned> hA(4096 * 4), hB(4096 * 4);
  std::vector<float> hX(65536);
  srand(3);
  auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); return u >> 16; };
  for (size_t i = 0; i < hA.size(); ++i) {
    hA[i] = bf((float)rand() / RAND_MAX - 0.5f) | (bf((float)rand() / RAND_MAX - 0.5f) << 16);
    hB[i] = bf((float)rand() / RAND_MAX - 0.5f) | (bf((float)rand() / RAND_MAX - 0.5f) << 16);
  }
  for (float& v : hX) v = (float)rand() / RAND_MAX * 4.0f - 2.0f;
  u32x4 *dA, *dB; float* dX; unsigned* dBad;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dX, hX.size() * 4); hipMalloc(&dBad, (size_t)1024 * 1024 * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
  const struct { const char* what; int blocks, threads, hammer; } shapes[] = {
      {"ONE wave per SIMD, no hammer (256 x 4 waves)", 256, 256, 0},
      {"TWO victims per SIMD, no hammer (256 x 8 waves)", 256, 512, 0},
      {"1 victim + 1 MFMA hammer per SIMD (256 x 8 waves)", 256, 512, 1},
      {"1 victim + 1 TRANS hammer per SIMD (256 x 8 waves)", 256, 512, 2},
      {"2 victims + 2 TRANS hammers per SIMD (256 x 16)", 256, 1024, 2},
      {"4 victims per SIMD, no hammer (256 x 16 waves)", 256, 1024, 0}};
  for (const auto& s : shapes) {
    printf("%-52s packed results that differ from the scalar ones (of %llu):  nothing in front %llu   1 own MFMA %llu   4 own MFMAs %llu   4 own v_rsq / v_rcp %llu\n",
           s.what, 6ull * 2000ull * s.blocks * s.threads / (s.hammer ? 2 : 1), run<0>(dA, dB, dX, dBad, s.blocks, s.threads, s.hammer),
           run<1>(dA, dB, dX, dBad, s.blocks, s.threads, s.hammer), run<2>(dA, dB, dX, dBad, s.blocks, s.threads, s.hammer),
           run<3>(dA, dB, dX, dBad, s.blocks, s.threads, s.hammer));
  }
  return 0;
}
