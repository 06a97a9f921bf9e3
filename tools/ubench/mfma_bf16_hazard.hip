// Which instruction distance does v_mfma_f32_16x16x32_bf16 need on gfx950?  Hand-placed sequences (inline asm, fixed registers, so
// that neither the scheduler nor the hazard recogniser of the compiler is involved) at full occupancy, checked against the same
// sequence with 64 idle cycles everywhere.
//   RAW k : mfma D <- A B ; k wait states ; v_mov out <- D          (a result read k wait states behind its MFMA)
//   RAW2 k: mfma D1 ; mfma D2 ; k wait states ; v_mov out <- D1     (... with a second MFMA in between)
//   WAR k : mfma D <- A B ; k wait states ; v_mov A.x <- junk ; 64 cycles ; read D   (a source overwritten behind its MFMA)
//   WAW k : mfma D ; k wait states ; v_mov D[0] <- 7.0 ; 64 cycles ; read D[0]      (a VALU write to the destination: who wins)
// hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_bf16_hazard.hip -o tools/ubench/mfma_bf16_hazard && tools/ubench/mfma_bf16_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define NOP16 "s_nop 15\n\t"
#define NOP64 NOP16 NOP16 NOP16 NOP16

template <int MODE, int K>
__global__ void kern(const u32x4* __restrict__ A, const u32x4* __restrict__ B, float* __restrict__ out, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    u32x4 a = A[(gid + it * 64) & 4095], b = B[(gid * 3 + it) & 4095];
    float r0;
    if (MODE == 0) {          // RAW
      asm volatile("v_mfma_f32_16x16x32_bf16 v[40:43], %1, %2, 0\n\t"
                   ".rept %3\n\ts_nop 0\n\t.endr\n\t"
                   "v_mov_b32 %0, v40\n\t" NOP64
                   : "=v"(r0) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43");
    } else if (MODE == 1) {   // RAW with a second MFMA in between
      asm volatile("v_mfma_f32_16x16x32_bf16 v[40:43], %1, %2, 0\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[44:47], %2, %1, 0\n\t"
                   ".rept %3\n\ts_nop 0\n\t.endr\n\t"
                   "v_mov_b32 %0, v40\n\t" NOP64
                   : "=v"(r0) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    } else if (MODE == 2) {   // WAR on a source
      asm volatile("v_mov_b32 v48, %1\n\tv_mov_b32 v49, %2\n\tv_mov_b32 v50, %3\n\tv_mov_b32 v51, %4\n\t" NOP16
                   "v_mfma_f32_16x16x32_bf16 v[40:43], v[48:51], %5, 0\n\t"
                   ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                   "v_mov_b32 v48, 0x7fc00000\n\tv_mov_b32 v49, 0x7fc00000\n\tv_mov_b32 v50, 0x7fc00000\n\tv_mov_b32 v51, 0x7fc00000\n\t" NOP64
                   "v_mov_b32 %0, v40\n\t"
                   : "=v"(r0) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b), "n"(K)
                   : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51");
    } else if (MODE == 7) {   // RAW with a PACKED reader: v_pk_mul_f32 reads the 64-bit pair D[0:1]
      float r1;
      asm volatile("v_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 1.0\n\t" NOP16
                   "v_mfma_f32_16x16x32_bf16 v[40:43], %2, %3, 0\n\t"
                   ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                   "v_pk_mul_f32 v[54:55], v[40:41], v[52:53]\n\t" NOP64
                   "v_mov_b32 %0, v54\n\tv_mov_b32 %1, v55\n\t"
                   : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43", "v52", "v53", "v54", "v55");
      r0 += 3.0f * r1;
    } else if (MODE == 8) {   // RAW with a packed reader of the UPPER pair D[2:3]
      float r1;
      asm volatile("v_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 1.0\n\t" NOP16
                   "v_mfma_f32_16x16x32_bf16 v[40:43], %2, %3, 0\n\t"
                   ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                   "v_pk_mul_f32 v[54:55], v[42:43], v[52:53]\n\t" NOP64
                   "v_mov_b32 %0, v54\n\tv_mov_b32 %1, v55\n\t"
                   : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43", "v52", "v53", "v54", "v55");
      r0 += 3.0f * r1;
    } else if (MODE == 9) {   // RAW through a copy: v_mov t <- D[3] after k wait states, v_pk_mul reads {t, D[2]} right behind it
      float r1;
      asm volatile("v_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 1.0\n\t" NOP16
                   "v_mfma_f32_16x16x32_bf16 v[40:43], %2, %3, 0\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[44:47], %3, %2, 0\n\t"
                   ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                   "v_mov_b32 v56, v40\n\tv_mov_b32 v57, v44\n\t"
                   "v_pk_mul_f32 v[54:55], v[56:57], v[52:53]\n\t" NOP64
                   "v_mov_b32 %0, v54\n\tv_mov_b32 %1, v55\n\t"
                   : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "n"(K)
                   : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v52", "v53", "v54", "v55", "v56", "v57");
      r0 += 3.0f * r1;
    } else if (MODE == 10) {  // VALU write of the A source registers -> MFMA read, K wait states in between
      asm volatile("v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t" NOP16
                   "v_mov_b32 v48, %1\n\tv_mov_b32 v49, %2\n\tv_mov_b32 v50, %3\n\tv_mov_b32 v51, %4\n\t"
                   ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[40:43], v[48:51], %5, 0\n\t" NOP64
                   "v_mov_b32 %0, v40\n\t"
                   : "=v"(r0) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b), "n"(K)
                   : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51");
    } else if (MODE == 11) {  // ... with v_perm_b32 as the writer (the kernel's packing instruction)
      asm volatile("v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0x03020100\n\t" NOP16
                   "v_perm_b32 v48, %1, %1, v52\n\tv_perm_b32 v49, %2, %2, v52\n\tv_perm_b32 v50, %3, %3, v52\n\tv_perm_b32 v51, %4, %4, v52\n\t"
                   ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[40:43], v[48:51], %5, 0\n\t" NOP64
                   "v_mov_b32 %0, v40\n\t"
                   : "=v"(r0) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b), "n"(K)
                   : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51", "v52");
    } else if (MODE == 12) {  // scalar read of the LAST result register D[3]
      asm volatile("v_mfma_f32_16x16x32_bf16 v[40:43], %1, %2, 0\n\t"
                   ".rept %3\n\ts_nop 0\n\t.endr\n\t"
                   "v_mov_b32 %0, v43\n\t" NOP64
                   : "=v"(r0) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43");
    } else if (MODE == 4) {   // destination = the A source registers (what the register allocator may choose), K unused
      asm volatile("v_mov_b32 v48, %1\n\tv_mov_b32 v49, %2\n\tv_mov_b32 v50, %3\n\tv_mov_b32 v51, %4\n\t" NOP16
                   "v_mfma_f32_16x16x32_bf16 v[48:51], v[48:51], %5, 0\n\t" NOP64
                   "v_mov_b32 %0, v48\n\t"
                   : "=v"(r0) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b), "n"(K)
                   : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51");
    } else if (MODE == 5) {   // destination = the B source registers
      asm volatile("v_mov_b32 v48, %1\n\tv_mov_b32 v49, %2\n\tv_mov_b32 v50, %3\n\tv_mov_b32 v51, %4\n\t" NOP16
                   "v_mfma_f32_16x16x32_bf16 v[48:51], %5, v[48:51], 0\n\t" NOP64
                   "v_mov_b32 %0, v48\n\t"
                   : "=v"(r0) : "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(a), "n"(K)
                   : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51");
    } else if (MODE == 6) {   // back-to-back MFMAs into the SAME destination, K wait states apart, the first one's value must not survive
      asm volatile("v_mfma_f32_16x16x32_bf16 v[40:43], %2, %1, 0\n\t"
                   ".rept %3\n\ts_nop 0\n\t.endr\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[40:43], %1, %2, 0\n\t" NOP64
                   "v_mov_b32 %0, v40\n\t"
                   : "=v"(r0) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43");
    } else {                  // WAW: VALU write to the destination behind the MFMA
      asm volatile("v_mfma_f32_16x16x32_bf16 v[40:43], %1, %2, 0\n\t"
                   ".rept %3\n\ts_nop 0\n\t.endr\n\t"
                   "v_mov_b32 v40, 7.0\n\t" NOP64
                   "v_mov_b32 %0, v40\n\t"
                   : "=v"(r0) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43");
    }
    acc += r0;
  }
  out[gid] = acc;
}

template <int MODE, int K>
static double run(const u32x4* dA, const u32x4* dB, float* dOut, std::vector<float>& h, int blocks) {
  hipLaunchKernelGGL((kern<MODE, K>), dim3(blocks), dim3(256), 0, 0, dA, dB, dOut, 64);
  hipMemcpy(h.data(), dOut, h.size() * 4, hipMemcpyDeviceToHost);
  double s = 0;
  for (float v : h) s += (double)v;
  return s;
}

template <int MODE>
static void sweep(const char* name, const u32x4* dA, const u32x4* dB, float* dOut, std::vector<float>& h, int blocks) {
  std::vector<float> ref(h.size());
  run<MODE, 64>(dA, dB, dOut, ref, blocks);
  auto bad = [&](std::vector<float>& g) { size_t n = 0; for (size_t i = 0; i < g.size(); ++i) n += (g[i] != ref[i]) && !(g[i] != g[i] && ref[i] != ref[i]); return n; };
  printf("%-5s wait states -> lanes that differ from the 64-wait-state run (of %zu):", name, h.size());
#define ONE(K) { run<MODE, K>(dA, dB, dOut, h, blocks); printf("  %d:%zu", K, bad(h)); }
  ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(14) ONE(16) ONE(18) ONE(20) ONE(24) ONE(32)
#undef ONE
  printf("\n");
}

int main() {
  const int blocks = 256 * 8;          // 2048 workgroups of 4 waves: every SIMD busy with several waves
  std::vector<unsigned> hA(4096 * 4), hB(4096 * 4);
  srand(3);
  auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); return u >> 16; };
  for (size_t i = 0; i < hA.size(); ++i) {
    hA[i] = bf((float)rand() / RAND_MAX - 0.5f) | (bf((float)rand() / RAND_MAX - 0.5f) << 16);
    hB[i] = bf((float)rand() / RAND_MAX - 0.5f) | (bf((float)rand() / RAND_MAX - 0.5f) << 16);
  }
  u32x4 *dA, *dB; float* dOut;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dOut, (size_t)blocks * 256 * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> h((size_t)blocks * 256);
  sweep<0>("RAW", dA, dB, dOut, h, blocks);
  sweep<1>("RAW2", dA, dB, dOut, h, blocks);
  sweep<2>("WAR", dA, dB, dOut, h, blocks);
  sweep<3>("WAW", dA, dB, dOut, h, blocks);
  {   // in-place destinations against the separate-destination result (MODE 0 with 64 wait states)
    std::vector<float> ref(h.size());
    run<0, 64>(dA, dB, dOut, ref, blocks);
    run<4, 0>(dA, dB, dOut, h, blocks);
    size_t n = 0; for (size_t i = 0; i < h.size(); ++i) n += h[i] != ref[i];
    printf("D = A registers: %zu lanes differ from the separate-destination result\n", n);
    // (MODE 5 swaps the operand roles: compare with the swapped reference)
    std::vector<float> ref5(h.size());
    hipLaunchKernelGGL((kern<1, 64>), dim3(blocks), dim3(256), 0, 0, dA, dB, dOut, 64);      // MODE 1 also computes mfma(b, a) into v[44:47] but returns D1
    run<5, 0>(dA, dB, dOut, h, blocks);
    run<5, 1>(dA, dB, dOut, ref5, blocks);
    n = 0; for (size_t i = 0; i < h.size(); ++i) n += h[i] != ref5[i];
    printf("D = B registers: run-to-run %zu lanes differ\n", n);
  }
  sweep<6>("WAWmm", dA, dB, dOut, h, blocks);
  sweep<7>("RAWpk", dA, dB, dOut, h, blocks);
  sweep<8>("RAWpkH", dA, dB, dOut, h, blocks);
  sweep<9>("RAWcp", dA, dB, dOut, h, blocks);
  sweep<10>("VtoA", dA, dB, dOut, h, blocks);
  sweep<11>("PtoA", dA, dB, dOut, h, blocks);
  sweep<12>("RAWd3", dA, dB, dOut, h, blocks);
  return 0;
}
