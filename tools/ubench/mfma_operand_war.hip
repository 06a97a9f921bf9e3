// Does v_mfma_f32_16x16x32_bf16 (gfx950) still need its 4-VGPR source operands AFTER it has issued?
//
// Round 5 found builds of amis_backward_mfma_kernel that returned wrong, run-to-run different gradients whenever two or more waves
// shared a SIMD (profiles/r05_bwd_scratch.txt).  Statically, what the failing builds have and the passing ones lack is a write -- a
// VALU instruction or the return of a scratch reload -- to the UPPER two registers of an MFMA's SrcA / SrcB tuple within one or two
// wait states behind the MFMA (tools/mfma_war_audit.py).  The compiler's hazard recogniser has no rule for that (sources are taken to
// be read at issue).  This probe tests the hardware directly, with hand-placed instructions (inline asm, fixed registers: neither the
// scheduler nor the recogniser is involved):
//
//     victim:      mfma D <- A, B ; K wait states ; v_mov <half of A or B> <- junk ; 64 idle cycles ; read D
//     hammer:      every second wave of a SIMD (waves 4-7, 12-15 of a workgroup) issues nothing but back-to-back independent MFMAs, so
//                  that at the victim's issue the matrix pipe of the SIMD is busy with ANOTHER wave's MFMAs (with one wave per SIMD --
//                  256-thread workgroups, one per CU -- it never is); the victim waves also run a burst of their own in front of
//                  the victim MFMA (a queue of the wave's own MFMAs)
//
// D is compared with the same sequence at K = 64.  hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_operand_war.hip -o tools/ubench/mfma_operand_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define NOP16 "s_nop 15\n\t"
#define NOP64 NOP16 NOP16 NOP16 NOP16
#define BG4 "v_mfma_f32_16x16x32_bf16 v[60:63], %2, %3, 0\n\tv_mfma_f32_16x16x32_bf16 v[64:67], %3, %2, 0\n\t" \
            "v_mfma_f32_16x16x32_bf16 v[68:71], %2, %3, 0\n\tv_mfma_f32_16x16x32_bf16 v[72:75], %3, %2, 0\n\t"

// OPER 0: the victim tuple is SrcA, 1: SrcB.  HALF 0: registers 0-1 of the tuple are overwritten, 1: registers 2-3.
template <int OPER, int HALF, int K>
__global__ void kern(const u32x4* __restrict__ A, const u32x4* __restrict__ B, float* __restrict__ out, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const u32x4 a = A[gid & 4095], b = B[(gid * 3) & 4095];
  const u32x4 vict = OPER == 0 ? a : b, other = OPER == 0 ? b : a;
  float acc = 0.f;
  if ((threadIdx.x >> 8) & 1) {       // hammer wave: MFMAs only, for as long as the victims run
    for (int it = 0; it < 3 * iters; ++it)
      asm volatile(BG4 BG4 BG4 BG4 BG4 BG4 BG4 BG4 : "=v"(acc) : "v"(acc), "v"(a), "v"(b)
                   : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");
    out[gid] = 0.f;
    return;
  }
  for (int it = 0; it < iters; ++it) {
    float r0, r1;
    if (OPER == 0) {
      asm volatile("v_mov_b32 v48, %4\n\tv_mov_b32 v49, %5\n\tv_mov_b32 v50, %6\n\tv_mov_b32 v51, %7\n\t" NOP16
                   BG4 BG4
                   "v_mfma_f32_16x16x32_bf16 v[40:43], v[48:51], %3, 0\n\t"
                   ".rept %8\n\ts_nop 0\n\t.endr\n\t"
                   ".if %9\n\tv_mov_b32 v50, 0x7fc00000\n\tv_mov_b32 v51, 0x7fc00000\n\t.else\n\tv_mov_b32 v48, 0x7fc00000\n\tv_mov_b32 v49, 0x7fc00000\n\t.endif\n\t" NOP64
                   "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v43\n\t"
                   : "=v"(r0), "=v"(r1)
                   : "v"(a), "v"(other), "v"(vict.x), "v"(vict.y), "v"(vict.z), "v"(vict.w), "n"(K), "n"(HALF)
                   : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",
                     "v69", "v70", "v71", "v72", "v73", "v74", "v75");
    } else if (OPER == 2) {   // RAW: the result read K wait states behind the MFMA (HALF: D[3] / D[0] first), own burst in front or not
      asm volatile(NOP16
                   ".if %9\n\t" BG4 BG4 ".endif\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[40:43], %2, %3, 0\n\t"
                   ".rept %8\n\ts_nop 0\n\t.endr\n\t"
                   "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v43\n\t" NOP64
                   : "=v"(r0), "=v"(r1)
                   : "v"(a), "v"(other), "v"(vict.x), "v"(vict.y), "v"(vict.z), "v"(vict.w), "n"(K), "n"(HALF)
                   : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",
                     "v69", "v70", "v71", "v72", "v73", "v74", "v75");
    } else {
      asm volatile("v_mov_b32 v48, %4\n\tv_mov_b32 v49, %5\n\tv_mov_b32 v50, %6\n\tv_mov_b32 v51, %7\n\t" NOP16
                   BG4 BG4
                   "v_mfma_f32_16x16x32_bf16 v[40:43], %3, v[48:51], 0\n\t"
                   ".rept %8\n\ts_nop 0\n\t.endr\n\t"
                   ".if %9\n\tv_mov_b32 v50, 0x7fc00000\n\tv_mov_b32 v51, 0x7fc00000\n\t.else\n\tv_mov_b32 v48, 0x7fc00000\n\tv_mov_b32 v49, 0x7fc00000\n\t.endif\n\t" NOP64
                   "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v43\n\t"
                   : "=v"(r0), "=v"(r1)
                   : "v"(a), "v"(other), "v"(vict.x), "v"(vict.y), "v"(vict.z), "v"(vict.w), "n"(K), "n"(HALF)
                   : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",
                     "v69", "v70", "v71", "v72", "v73", "v74", "v75");
    }
    acc += r0 + 3.0f * r1;
  }
  out[gid] = acc;
}

template <int OPER, int HALF, int K>
static void run(const u32x4* dA, const u32x4* dB, float* dOut, std::vector<float>& h, int blocks, int threads) {
  hipLaunchKernelGGL((kern<OPER, HALF, K>), dim3(blocks), dim3(threads), 0, 0, dA, dB, dOut, 200);
  hipMemcpy(h.data(), dOut, (size_t)blocks * threads * 4, hipMemcpyDeviceToHost);
}

template <int OPER, int HALF>
static void sweep(const char* name, const u32x4* dA, const u32x4* dB, float* dOut, int blocks, int threads) {
  const size_t n = (size_t)blocks * threads;
  std::vector<float> ref(n), h(n);
  run<OPER, HALF, 64>(dA, dB, dOut, ref, blocks, threads);
  auto bad = [&]() { size_t c = 0; for (size_t i = 0; i < n; ++i) c += (h[i] != ref[i]) && !(h[i] != h[i] && ref[i] != ref[i]); return c; };
  printf("  %-26s", name);
#define ONE(K) { run<OPER, HALF, K>(dA, dB, dOut, h, blocks, threads); printf("  %d:%zu", K, bad()); }
  ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(14) ONE(16) ONE(24) ONE(32)
#undef ONE
  printf("\n");
}

int main() {
  std::vector<unsigned> hA(4096 * 4), hB(4096 * 4);
  srand(3);
  auto bf = [](float f) { unsigned u; memcpy(&u, &f, 4); return u >> 16; };
  for (size_t i = 0; i < hA.size(); ++i) {
    hA[i] = bf((float)rand() / RAND_MAX - 0.5f) | (bf((float)rand() / RAND_MAX - 0.5f) << 16);
    hB[i] = bf((float)rand() / RAND_MAX - 0.5f) | (bf((float)rand() / RAND_MAX - 0.5f) << 16);
  }
  u32x4 *dA, *dB; float* dOut;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dOut, (size_t)4096 * 1024 * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  const struct { const char* what; int blocks, threads; } shapes[] = {
      {"ONE wave per SIMD (256 workgroups x 4 waves, no hammer)", 256, 256},
      {"TWO waves per SIMD: 1 victim + 1 hammer (256 x 8 waves)", 256, 512},
      {"FOUR waves per SIMD: 2 victims + 2 hammers (256 x 16 waves)", 256, 1024},
      {"EIGHT waves per SIMD: 4 + 4 (512 x 16 waves)", 512, 1024}};
  for (const auto& s : shapes) {
    printf("%s -- lanes (of %d) whose D differs from the 64-wait-state run, by wait states between the MFMA and the overwrite:\n", s.what,
           s.blocks * s.threads);
    sweep<0, 0>("SrcA registers 0-1", dA, dB, dOut, s.blocks, s.threads);
    sweep<0, 1>("SrcA registers 2-3", dA, dB, dOut, s.blocks, s.threads);
    sweep<1, 0>("SrcB registers 0-1", dA, dB, dOut, s.blocks, s.threads);
    sweep<1, 1>("SrcB registers 2-3", dA, dB, dOut, s.blocks, s.threads);
    sweep<2, 0>("RAW, idle pipe in front", dA, dB, dOut, s.blocks, s.threads);
    sweep<2, 1>("RAW, 8 own MFMAs in front", dA, dB, dOut, s.blocks, s.threads);
  }
  return 0;
}
