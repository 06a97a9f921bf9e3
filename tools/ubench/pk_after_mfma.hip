// gfx950: a packed fp32 instruction issued in the shadow of the wave's OWN v_mfma_f32_16x16x32_bf16, several waves per SIMD.
//
// Follow-up of tools/ubench/pk_opsel_under_mfma.hip, which found wrong packed results only in the column "one MFMA of the wave's own
// directly in front" at >= 4 waves per SIMD.  Here every kernel issues ONE kind of VALU instruction four times (four destination
// pairs, the same sources) K wait states behind one MFMA that shares no register with them, and compares with scalar arithmetic:
//     plain    v_pk_mul_f32 D, A, B
//     hi_lo    v_pk_mul_f32 D, A, B op_sel_hi:[1,0]          (D.hi = A.hi * B.lo)
//     lo_hi    v_pk_mul_f32 D, A, B op_sel:[0,1]             (D.lo = A.lo * B.hi)
//     fma_sel  v_pk_fma_f32 D, A, B, C op_sel:[1,0,0]
//     scalar   v_mul_f32 d, a, b  (x4)                      (control: an unpacked VALU instruction in the same place)
// hipcc --offload-arch=gfx950 -O2 -w tools/ubench/pk_after_mfma.hip -o tools/ubench/pk_after_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define PK4(INS, MOD)                                                                                                              \
  INS " v[86:87], v[80:81], v[82:83]" MOD "\n\t" INS " v[88:89], v[80:81], v[82:83]" MOD "\n\t" INS " v[90:91], v[80:81], v[82:83]" MOD \
      "\n\t" INS " v[92:93], v[80:81], v[82:83]" MOD "\n\t"

// KIND 0 plain, 1 hi_lo, 2 lo_hi, 3 fma_sel, 4 scalar.  MF: 1 = one MFMA in front, 0 = none.  K: wait states between the two.
template <int KIND, int MF, int K>
__global__ void kern(const u32x4* __restrict__ A, const u32x4* __restrict__ B, const float* __restrict__ X, unsigned* __restrict__ bad, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const u32x4 a = A[gid & 4095], b = B[(gid * 3) & 4095];
  unsigned nbad = 0;
  float x0 = X[gid & 65535], x1 = X[(gid + 7) & 65535], y0 = X[(gid * 5 + 1) & 65535], y1 = X[(gid * 11 + 3) & 65535];
  float c0 = X[(gid * 13 + 5) & 65535], c1 = X[(gid * 17 + 9) & 65535];
  const int u0 = __builtin_amdgcn_readfirstlane(__float_as_int(X[(blockIdx.x * 7 + 1) & 65535])), u1 = __builtin_amdgcn_readfirstlane(__float_as_int(X[(blockIdx.x * 3 + 2) & 65535]));
  for (int it = 0; it < iters; ++it) {
    float r[8];
    asm volatile("s_mov_b32 s40, %19\n\ts_mov_b32 s41, %20\n\tv_mov_b32 v80, %8\n\tv_mov_b32 v81, %9\n\tv_mov_b32 v82, %10\n\tv_mov_b32 v83, %11\n\tv_mov_b32 v84, %12\n\tv_mov_b32 v85, %13\n\t"
                 "s_nop 7\n\ts_nop 7\n\t"
                 ".if %16 == 1\n\t v_mfma_f32_16x16x32_bf16 v[60:63], %14, %15, 0\n\t.endif\n\t"
                 ".if %16 == 2\n\t v_mfma_f32_16x16x4_f32 v[60:63], v80, v82, 0\n\t.endif\n\t"
                 ".rept %17\n\ts_nop 0\n\t.endr\n\t"
                 ".if %18 == 0\n\t" PK4("v_pk_mul_f32", "") ".endif\n\t"
                 ".if %18 == 1\n\t" PK4("v_pk_mul_f32", " op_sel_hi:[1,0]") ".endif\n\t"
                 ".if %18 == 2\n\t" PK4("v_pk_mul_f32", " op_sel:[0,1]") ".endif\n\t"
                 ".if %18 == 3\n\t"
                 "v_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0]\n\t"
                 "v_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0]\n\t"
                 ".endif\n\t"
                 ".if %18 == 5\n\t" PK4("v_pk_mul_f32", " op_sel:[1,0]") ".endif\n\t"
                 ".if %18 == 6\n\t"
                 "v_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel:[0,1,0]\n\t"
                 "v_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel:[0,1,0]\n\t"
                 ".endif\n\t"
                 ".if %18 == 7\n\t" PK4("v_pk_add_f32", " op_sel:[0,1]") ".endif\n\t"
                 ".if %18 == 10\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel:[0,0,1]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel:[0,0,1]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel:[0,0,1]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel:[0,0,1]\n\t.endif\n\t"
                 ".if %18 == 11\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,1,0]\n\t.endif\n\t"
                 ".if %18 == 12\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,1]\n\t.endif\n\t"
                 ".if %18 == 13\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1]\n\t.endif\n\t"
                 ".if %18 == 14\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\t.endif\n\t"
                 ".if %18 == 15\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel:[1,0,0] op_sel_hi:[1,0,1]\n\t.endif\n\t"
                 ".if %18 == 16\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,0]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,0]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,0]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,0]\n\t.endif\n\t"
                 ".if %18 == 17\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel:[1,1,0]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel:[1,1,0]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel:[1,1,0]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel:[1,1,0]\n\t.endif\n\t"
                 ".if %18 == 20\n\t" "v_pk_mul_f32 v[86:87], s[40:41], v[82:83]\n\tv_pk_mul_f32 v[88:89], s[40:41], v[82:83]\n\tv_pk_mul_f32 v[90:91], s[40:41], v[82:83]\n\tv_pk_mul_f32 v[92:93], s[40:41], v[82:83]\n\t" ".endif\n\t"
                 ".if %18 == 21\n\t" "v_pk_fma_f32 v[86:87], s[40:41], v[82:83], v[84:85]\n\tv_pk_fma_f32 v[88:89], s[40:41], v[82:83], v[84:85]\n\tv_pk_fma_f32 v[90:91], s[40:41], v[82:83], v[84:85]\n\tv_pk_fma_f32 v[92:93], s[40:41], v[82:83], v[84:85]\n\t" ".endif\n\t"
                 ".if %18 == 22\n\t" "v_pk_fma_f32 v[86:87], s[40:41], v[82:83], v[84:85] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[88:89], s[40:41], v[82:83], v[84:85] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[90:91], s[40:41], v[82:83], v[84:85] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 v[92:93], s[40:41], v[82:83], v[84:85] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\t" ".endif\n\t"
                 ".if %18 == 23\n\t" "v_pk_fma_f32 v[86:87], v[80:81], s[40:41], v[84:85] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[88:89], v[80:81], s[40:41], v[84:85] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[90:91], v[80:81], s[40:41], v[84:85] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[92:93], v[80:81], s[40:41], v[84:85] op_sel_hi:[1,0,1]\n\t" ".endif\n\t"
                 ".if %18 == 30\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t.endif\n\t"
                 ".if %18 == 31\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t.endif\n\t"
                 ".if %18 == 32\n\tv_pk_mul_f32 v[86:87], v[80:81], v[82:83] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 v[88:89], v[80:81], v[82:83] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 v[90:91], v[80:81], v[82:83] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 v[92:93], v[80:81], v[82:83] neg_lo:[0,1] neg_hi:[0,1]\n\t.endif\n\t"
                 ".if %18 == 33\n\tv_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[82:83], v[84:85] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[82:83], v[84:85] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[82:83], v[84:85] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t.endif\n\t"
                 ".if %18 == 34\n\tv_pk_fma_f32 v[86:87], v[80:81], 2.0, v[84:85] op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 v[88:89], v[80:81], 2.0, v[84:85] op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 v[90:91], v[80:81], 2.0, v[84:85] op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 v[92:93], v[80:81], 2.0, v[84:85] op_sel_hi:[1,0,0]\n\t.endif\n\t"
                 ".if %18 == 35\n\tv_pk_fma_f32 v[86:87], v[80:81], v[80:81], v[84:85] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[88:89], v[80:81], v[80:81], v[84:85] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[90:91], v[80:81], v[80:81], v[84:85] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 v[92:93], v[80:81], v[80:81], v[84:85] op_sel_hi:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t.endif\n\t"
                 ".if %18 == 4\n\t"
                 "v_mul_f32 v86, v80, v82\n\tv_mul_f32 v87, v81, v83\n\tv_mul_f32 v88, v80, v82\n\tv_mul_f32 v89, v81, v83\n\t"
                 "v_mul_f32 v90, v80, v82\n\tv_mul_f32 v91, v81, v83\n\tv_mul_f32 v92, v80, v82\n\tv_mul_f32 v93, v81, v83\n\t"
                 ".endif\n\t"
                 "s_nop 7\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v86\n\tv_mov_b32 %1, v87\n\tv_mov_b32 %2, v88\n\tv_mov_b32 %3, v89\n\tv_mov_b32 %4, v90\n\tv_mov_b32 %5, v91\n\t"
                 "v_mov_b32 %6, v92\n\tv_mov_b32 %7, v93\n\t"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1), "v"(a), "v"(b), "n"(MF), "n"(K), "n"(KIND), "s"(u0), "s"(u1)
                 : "s40", "s41", "v60", "v61", "v62", "v63", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93");
    float e0, e1;
    if (KIND == 0 || KIND == 4) asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
    if (KIND == 1) asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
    if (KIND == 2) asm volatile("v_mul_f32 %0, %2, %5\n\tv_mul_f32 %1, %3, %5\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
    if (KIND == 5) asm volatile("v_mul_f32 %0, %3, %4\n\tv_mul_f32 %1, %3, %5\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
    if (KIND == 6) asm volatile("v_fma_f32 %0, %2, %5, %6\n\tv_fma_f32 %1, %3, %5, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 7) asm volatile("v_add_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %5\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
    if (KIND == 10) asm volatile("v_fma_f32 %0, %2, %4, %7\n\tv_fma_f32 %1, %3, %5, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 11) asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %6\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 12) asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %2, %5, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 13) asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %4, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 14) asm volatile("v_fma_f32 %0, %2, %4, %7\n\tv_fma_f32 %1, %3, %5, %6\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 15) asm volatile("v_fma_f32 %0, %3, %4, %6\n\tv_fma_f32 %1, %3, %4, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 16) asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %2, %5, %6\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 17) asm volatile("v_fma_f32 %0, %3, %5, %6\n\tv_fma_f32 %1, %3, %5, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 20) asm volatile("v_mul_f32 %0, %6, %4\n\tv_mul_f32 %1, %7, %5\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "s"(u0), "s"(u1));
    if (KIND == 21) asm volatile("v_fma_f32 %0, %8, %4, %6\n\tv_fma_f32 %1, %9, %5, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1), "s"(u0), "s"(u1));
    if (KIND == 22) asm volatile("v_fma_f32 %0, %8, %4, %7\n\tv_fma_f32 %1, %9, %5, %6\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1), "s"(u0), "s"(u1));
    if (KIND == 23) asm volatile("v_fma_f32 %0, %2, %8, %6\n\tv_fma_f32 %1, %3, %8, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1), "s"(u0), "s"(u1));
    if (KIND == 30) asm volatile("v_fma_f32 %0, %2, %4, -%6\n\tv_fma_f32 %1, %2, %5, -%6\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 31) asm volatile("v_fma_f32 %0, %2, %4, -%6\n\tv_fma_f32 %1, %3, %4, -%7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 32) asm volatile("v_mul_f32 %0, %2, -%4\n\tv_mul_f32 %1, %3, -%5\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 33) asm volatile("v_fma_f32 %0, -%2, %4, %6\n\tv_fma_f32 %1, -%3, %5, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 34) asm volatile("v_fma_f32 %0, %2, 2.0, %6\n\tv_fma_f32 %1, %3, 2.0, %6\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 35) asm volatile("v_fma_f32 %0, %2, %2, -%6\n\tv_fma_f32 %1, %2, %3, -%6\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
    if (KIND == 3) asm volatile("v_fma_f32 %0, %3, %4, %6\n\tv_fma_f32 %1, %3, %5, %7\n\ts_nop 1\n\t" : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(c0), "v"(c1));
#pragma unroll
    for (int k = 0; k < 4; ++k)
      nbad += (__float_as_uint(r[2 * k]) != __float_as_uint(e0)) + (__float_as_uint(r[2 * k + 1]) != __float_as_uint(e1));
    x0 = x0 * 1.0000001f + 0.25f; y1 = y1 * 0.9999999f - 0.125f;
  }
  bad[gid] = nbad;
}

template <int KIND, int MF, int K>
static unsigned long long run(const u32x4* dA, const u32x4* dB, const float* dX, unsigned* dBad, int blocks, int threads) {
  hipMemset(dBad, 0, (size_t)blocks * threads * 4);
  hipLaunchKernelGGL((kern<KIND, MF, K>), dim3(blocks), dim3(threads), 0, 0, dA, dB, dX, dBad, 4000);
  std::vector<unsigned> h((size_t)blocks * threads);
  hipMemcpy(h.data(), dBad, h.size() * 4, hipMemcpyDeviceToHost);
  unsigned long long s = 0;
  for (unsigned v : h) s += v;
  return s;
}

template <int KIND, int MF>
static void row(const char* name, const u32x4* dA, const u32x4* dB, const float* dX, unsigned* dBad, int blocks, int threads) {
  printf("  %-34s no MFMA: %-4llu behind the MFMA, wait states 0: %-6llu 2: %-6llu 4: %-6llu 8: %-6llu 12: %-8llu 14: %-8llu 16: %-8llu 18: %-8llu 20: %-8llu 24: %-8llu 32: %llu\n", name,
         run<KIND, 0, 0>(dA, dB, dX, dBad, blocks, threads), run<KIND, MF, 0>(dA, dB, dX, dBad, blocks, threads),
         run<KIND, MF, 2>(dA, dB, dX, dBad, blocks, threads), run<KIND, MF, 4>(dA, dB, dX, dBad, blocks, threads),
         run<KIND, MF, 8>(dA, dB, dX, dBad, blocks, threads), run<KIND, MF, 12>(dA, dB, dX, dBad, blocks, threads),
         run<KIND, MF, 14>(dA, dB, dX, dBad, blocks, threads), run<KIND, MF, 16>(dA, dB, dX, dBad, blocks, threads),
         run<KIND, MF, 18>(dA, dB, dX, dBad, blocks, threads), run<KIND, MF, 20>(dA, dB, dX, dBad, blocks, threads),
         run<KIND, MF, 24>(dA, dB, dX, dBad, blocks, threads), run<KIND, MF, 32>(dA, dB, dX, dBad, blocks, threads));
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
  const struct { const char* what; int blocks, threads; } shapes[] = {
      {"ONE wave per SIMD (256 x 4 waves)", 256, 256}, {"TWO waves per SIMD (256 x 8)", 256, 512}, {"FOUR waves per SIMD (256 x 16)", 256, 1024}};
  for (const auto& s : shapes) {
    printf("%s -- wrong results of %llu per cell:\n", s.what, 8ull * 4000ull * s.blocks * s.threads);
    row<0, 1>("plain", dA, dB, dX, dBad, s.blocks, s.threads);
    row<1, 1>("mul [..] hi:[1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<2, 1>("mul sel:[0,1]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<5, 1>("mul sel:[1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<3, 1>("fma sel:[1,0,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<6, 1>("fma sel:[0,1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<7, 1>("add sel:[0,1]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<10, 1>("fma sel:[0,0,1]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<11, 1>("fma sel_hi:[1,1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<12, 1>("fma sel_hi:[0,1,1]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<13, 1>("fma sel_hi:[1,0,1]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<14, 1>("fma sel:[0,0,1] sel_hi:[1,1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<15, 1>("fma sel:[1,0,0] sel_hi:[1,0,1]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<16, 1>("fma sel_hi:[0,1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<17, 1>("fma sel:[1,1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<20, 1>("mul s[..], v", dA, dB, dX, dBad, s.blocks, s.threads);
    row<21, 1>("fma s[..], v, v", dA, dB, dX, dBad, s.blocks, s.threads);
    row<22, 1>("fma s[..] sel:[0,0,1] hi:[1,1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<23, 1>("fma v, s[..], v hi:[1,0,1]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<30, 1>("fma hi:[0,1,0] neg c", dA, dB, dX, dBad, s.blocks, s.threads);
    row<31, 1>("fma hi:[1,0,1] neg c", dA, dB, dX, dBad, s.blocks, s.threads);
    row<32, 1>("mul neg b", dA, dB, dX, dBad, s.blocks, s.threads);
    row<33, 1>("fma neg a", dA, dB, dX, dBad, s.blocks, s.threads);
    row<34, 1>("fma v, 2.0, v hi:[1,0,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<35, 1>("fma hi:[0,1,0]", dA, dB, dX, dBad, s.blocks, s.threads);
    row<4, 1>("scalar", dA, dB, dX, dBad, s.blocks, s.threads);
    row<2, 2>("mul sel:[0,1] behind the FP32 MFMA 16x16x4", dA, dB, dX, dBad, s.blocks, s.threads);
  }
  return 0;
}
