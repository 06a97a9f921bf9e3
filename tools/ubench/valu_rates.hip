// Micro-benchmark: issue rate of fp32 VALU op classes on gfx950 as a function of waves per SIMD.
// 64 instructions per loop iteration (8 independent register chains x 8), so loop overhead is < 5 %.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define X8(S) S S S S S S S S

template <int MODE>
__global__ void k(float* out, long long* clk, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  const f2 A = {a, a}, Bv = {b, b};
  float sa = a;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#define SC(OPSTR) asm volatile(X8(OPSTR " %0, %0, %8, %9\n" OPSTR " %1, %1, %8, %9\n" OPSTR " %2, %2, %8, %9\n" OPSTR " %3, %3, %8, %9\n" \
                                  OPSTR " %4, %4, %8, %9\n" OPSTR " %5, %5, %8, %9\n" OPSTR " %6, %6, %8, %9\n" OPSTR " %7, %7, %8, %9\n") \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b))
#define SC2(OPSTR) asm volatile(X8(OPSTR " %0, %0, %8\n" OPSTR " %1, %1, %8\n" OPSTR " %2, %2, %8\n" OPSTR " %3, %3, %8\n" \
                                   OPSTR " %4, %4, %8\n" OPSTR " %5, %5, %8\n" OPSTR " %6, %6, %8\n" OPSTR " %7, %7, %8\n") \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a))
#define SC1(OPSTR) asm volatile(X8(OPSTR " %0, %0\n" OPSTR " %1, %1\n" OPSTR " %2, %2\n" OPSTR " %3, %3\n" \
                                   OPSTR " %4, %4\n" OPSTR " %5, %5\n" OPSTR " %6, %6\n" OPSTR " %7, %7\n") \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7))
#define PK(OPSTR) asm volatile(X8(OPSTR " %0, %0, %8, %9\n" OPSTR " %1, %1, %8, %9\n" OPSTR " %2, %2, %8, %9\n" OPSTR " %3, %3, %8, %9\n" \
                                  OPSTR " %4, %4, %8, %9\n" OPSTR " %5, %5, %8, %9\n" OPSTR " %6, %6, %8, %9\n" OPSTR " %7, %7, %8, %9\n") \
    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(A), "v"(Bv))
#define PK2(OPSTR) asm volatile(X8(OPSTR " %0, %0, %8\n" OPSTR " %1, %1, %8\n" OPSTR " %2, %2, %8\n" OPSTR " %3, %3, %8\n" \
                                   OPSTR " %4, %4, %8\n" OPSTR " %5, %5, %8\n" OPSTR " %6, %6, %8\n" OPSTR " %7, %7, %8\n") \
    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(A))
    if (MODE == 0) SC("v_fma_f32");
    else if (MODE == 1) PK("v_pk_fma_f32");
    else if (MODE == 2) SC1("v_rcp_f32");
    else if (MODE == 3) SC1("v_sqrt_f32");
    else if (MODE == 4) SC2("v_max_f32");
    else if (MODE == 5) SC2("v_mul_f32");
    else if (MODE == 6) PK2("v_pk_mul_f32");
    else if (MODE == 7) PK2("v_pk_add_f32");
    else if (MODE == 8) SC2("v_fmac_f32");
    else if (MODE == 9) {   // fma with an SGPR operand
      asm volatile(X8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(sa), "v"(b));
    } else if (MODE == 10) {   // 1 rcp : 7 fma
      asm volatile(X8("v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 11) {   // 2 trans : 6 fma (rcp + sqrt)
      asm volatile(X8("v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_sqrt_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 12) {   // 2 trans : 6 pk_fma
      asm volatile(X8("v_rcp_f32 %8, %8\n v_pk_fma_f32 %1, %1, %9, %10\n v_pk_fma_f32 %2, %2, %9, %10\n v_pk_fma_f32 %3, %3, %9, %10\n"
                      "v_sqrt_f32 %8, %8\n v_pk_fma_f32 %5, %5, %9, %10\n v_pk_fma_f32 %6, %6, %9, %10\n v_pk_fma_f32 %7, %7, %9, %10\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(x0) : "v"(A), "v"(Bv));
    } else if (MODE == 13) {   // v_add_f32 dpp row_ror
      asm volatile(X8("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                      "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 14) SC2("v_min_f32");
    else if (MODE == 15) SC1("v_rsq_f32");
    else if (MODE == 16) SC("v_med3_f32");
    else if (MODE == 17) SC("v_max3_f32");
    else if (MODE == 18) SC2("v_add_f32");
    else if (MODE == 19) {   // v_cmp_gt + v_cndmask pairs (what `x > t ? x : t` compiles to)
      asm volatile(X8("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %8, %1, vcc\n"
                      "v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %8, %2, vcc\n v_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %8, %3, vcc\n")
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd, double flops_per_instr) {
  float* out; long long* clk;
  const int blocks = 256 * waves_per_simd, threads = 256, iters = 3000;
  hipMalloc(&out, blocks * threads * sizeof(float)); hipMalloc(&clk, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(out, clk, 50, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, clk, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  double per_simd = (double)waves_per_simd * iters * 64;   // wave-instructions issued on each SIMD
  printf("%-26s w/SIMD=%d %8.3f ms  %6.2f ns/instr/SIMD  clock64/instr %6.2f  %7.1f TFLOP/s\n", name, waves_per_simd, ms,
         ms * 1e6 / per_simd, (double)c / per_simd, flops_per_instr * 64 * per_simd * 1024 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int w : {2, 8}) {
    run<0>("v_fma_f32", w, 2); run<9>("v_fma_f32 (sgpr src)", w, 2); run<8>("v_fmac_f32", w, 2); run<5>("v_mul_f32", w, 1);
    run<1>("v_pk_fma_f32", w, 4); run<6>("v_pk_mul_f32", w, 2); run<7>("v_pk_add_f32", w, 2);
    run<4>("v_max_f32", w, 1); run<14>("v_min_f32", w, 1); run<13>("v_add_f32_dpp", w, 1);
    run<2>("v_rcp_f32", w, 1); run<3>("v_sqrt_f32", w, 1); run<15>("v_rsq_f32", w, 1);
    run<16>("v_med3_f32", w, 1); run<17>("v_max3_f32", w, 1); run<18>("v_add_f32", w, 1); run<19>("v_cmp_gt+v_cndmask (x1)", w, 1);
    run<10>("1 rcp : 7 fma", w, 1.875); run<11>("2 trans : 6 fma", w, 1.75); run<12>("2 trans : 6 pk_fma", w, 3.25);
    printf("\n");
  }
  return 0;
}
