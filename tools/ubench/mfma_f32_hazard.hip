// The same hand-placed distance checks as mfma_bf16_hazard.hip for the instruction the shipped kernels use, v_mfma_f32_16x16x4_f32
// (8 passes): result reads of D[0] / D[3] / the packed upper pair, and a VALU-written source, K wait states apart, at full occupancy,
// against the same sequence with 64 wait states.  The compiler's hazard recogniser uses 11 wait states for the reads and 2 for the source.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define NOP16 "s_nop 15\n\t"
#define NOP64 NOP16 NOP16 NOP16 NOP16

template <int MODE, int K>
__global__ void kern(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float a = A[(gid + it * 64) & 4095], b = B[(gid * 3 + it) & 4095];
    float r0, r1 = 0.f;
    if (MODE == 0) {
      asm volatile("v_mfma_f32_16x16x4_f32 v[40:43], %1, %2, 0\n\t.rept %3\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %0, v40\n\t" NOP64
                   : "=v"(r0) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43");
    } else if (MODE == 1) {
      asm volatile("v_mfma_f32_16x16x4_f32 v[40:43], %1, %2, 0\n\t.rept %3\n\ts_nop 0\n\t.endr\n\tv_mov_b32 %0, v43\n\t" NOP64
                   : "=v"(r0) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43");
    } else if (MODE == 2) {
      asm volatile("v_mov_b32 v52, 1.0\n\tv_mov_b32 v53, 1.0\n\t" NOP16
                   "v_mfma_f32_16x16x4_f32 v[40:43], %2, %3, 0\n\t.rept %4\n\ts_nop 0\n\t.endr\n\t"
                   "v_pk_mul_f32 v[54:55], v[42:43], v[52:53]\n\t" NOP64 "v_mov_b32 %0, v54\n\tv_mov_b32 %1, v55\n\t"
                   : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43", "v52", "v53", "v54", "v55");
    } else {
      asm volatile("v_mov_b32 v48, 0\n\t" NOP16 "v_mov_b32 v48, %1\n\t.rept %3\n\ts_nop 0\n\t.endr\n\t"
                   "v_mfma_f32_16x16x4_f32 v[40:43], v48, %2, 0\n\t" NOP64 "v_mov_b32 %0, v40\n\t"
                   : "=v"(r0) : "v"(a), "v"(b), "n"(K) : "v40", "v41", "v42", "v43", "v48");
    }
    acc += r0 + 3.0f * r1;
  }
  out[gid] = acc;
}

template <int MODE, int K>
static void run(const float* dA, const float* dB, float* dOut, std::vector<float>& h, int blocks) {
  hipLaunchKernelGGL((kern<MODE, K>), dim3(blocks), dim3(256), 0, 0, dA, dB, dOut, 64);
  (void)hipMemcpy(h.data(), dOut, h.size() * 4, hipMemcpyDeviceToHost);
}

template <int MODE>
static void sweep(const char* name, const float* dA, const float* dB, float* dOut, std::vector<float>& h, int blocks) {
  std::vector<float> ref(h.size());
  run<MODE, 64>(dA, dB, dOut, ref, blocks);
  auto bad = [&](std::vector<float>& g) { size_t n = 0; for (size_t i = 0; i < g.size(); ++i) n += (g[i] != ref[i]) && !(g[i] != g[i] && ref[i] != ref[i]); return n; };
  printf("%-6s wait states -> lanes that differ from the 64-wait-state run (of %zu):", name, h.size());
#define ONE(K) { run<MODE, K>(dA, dB, dOut, h, blocks); printf("  %d:%zu", K, bad(h)); }
  ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(6) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(16) ONE(20) ONE(24) ONE(32)
#undef ONE
  printf("\n");
}

int main() {
  const int blocks = 256 * 8;
  std::vector<float> hA(4096), hB(4096);
  srand(3);
  for (size_t i = 0; i < hA.size(); ++i) { hA[i] = (float)rand() / (float)RAND_MAX - 0.5f; hB[i] = (float)rand() / (float)RAND_MAX - 0.5f; }
  float *dA, *dB, *dOut;
  (void)hipMalloc(&dA, 4096 * 4); (void)hipMalloc(&dB, 4096 * 4); (void)hipMalloc(&dOut, (size_t)blocks * 256 * 4);
  (void)hipMemcpy(dA, hA.data(), 4096 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB.data(), 4096 * 4, hipMemcpyHostToDevice);
  std::vector<float> h((size_t)blocks * 256);
  sweep<0>("RAWd0", dA, dB, dOut, h, blocks);
  sweep<1>("RAWd3", dA, dB, dOut, h, blocks);
  sweep<2>("RAWpkH", dA, dB, dOut, h, blocks);
  sweep<3>("VtoA", dA, dB, dOut, h, blocks);
  return 0;
}
