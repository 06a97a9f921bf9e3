// bf16x3-split projection vs the fp32 MFMA vs fp64: is ONE v_mfma_f32_16x16x32_bf16 on split operands as accurate as
// v_mfma_f32_16x16x4_f32?   hipcc --offload-arch=gfx950 -O3 -I epro-pnp_amd/csrc -I include tools/ubench/bf16_split_mfma.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct Dword4 { unsigned x, y, z, w; };
__device__ unsigned f32_bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
__device__ float bits_f32(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
__device__ unsigned bf16_rne(float f) { unsigned u = f32_bits(f); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; }
__device__ void split3(float a, unsigned& a1, unsigned& a2, unsigned& a3) {
  a1 = bf16_rne(a); const float r1 = a - bits_f32(a1 << 16);
  a2 = bf16_rne(r1); const float r2 = r1 - bits_f32(a2 << 16);
  a3 = bf16_rne(r2);
}
__device__ Dword4 split_a4(float a) { unsigned a1, a2, a3; split3(a, a1, a2, a3); const unsigned w0 = a1 | (a2 << 16); return Dword4{w0, w0, w0, a3 | (a3 << 16)}; }
__device__ Dword4 split_b4(float b) { unsigned b1, b2, b3; split3(b, b1, b2, b3); return Dword4{b1 | (b1 << 16), b2 | (b2 << 16), b3 | (b3 << 16), b1 | (b2 << 16)}; }

// A (16 x 4) row-major, B (4 x 16) row-major -> D32, Dsplit (16 x 16)
__global__ void k(const float* A, const float* B, float* D32, float* Ds, int mode) {
  const int lane = threadIdx.x, col = lane & 15, kk = lane >> 4;
  const float a = A[col * 4 + kk], b = B[kk * 16 + col];
  floatx4 z = {0, 0, 0, 0};
  floatx4 d32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, z, 0, 0, 0);
  Dword4 sa = split_a4(a), sb = split_b4(b);
  if (mode == 1) { sa = Dword4{sa.x, 0, 0, 0}; sb = Dword4{sb.x, 0, 0, 0}; }      // only a1b1 + a2b1: checks the element pairing
  floatx4 ds = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, sa), __builtin_bit_cast(bf16x8_t, sb), z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) { D32[(4 * kk + r) * 16 + col] = d32[r]; Ds[(4 * kk + r) * 16 + col] = ds[r]; }
}

int main() {
  float hA[64], hB[64], h32[256], hs[256];
  srand(1);
  double worst32 = 0, worsts = 0;
  for (int trial = 0; trial < 200; ++trial) {
    for (int i = 0; i < 64; ++i) { hA[i] = 800.f * ((float)rand() / RAND_MAX - 0.5f); hB[i] = (i / 16 == 3) ? 1.0f : ((float)rand() / RAND_MAX - 0.5f); }
    float *dA, *dB, *d32, *ds;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&d32, 1024); hipMalloc(&ds, 1024);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d32, ds, 0);
    hipMemcpy(h32, d32, 1024, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, 1024, hipMemcpyDeviceToHost);
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        double ref = 0, mag = 0;
        for (int kq = 0; kq < 4; ++kq) { ref += (double)hA[m * 4 + kq] * hB[kq * 16 + n]; mag += fabs((double)hA[m * 4 + kq] * hB[kq * 16 + n]); }
        worst32 = fmax(worst32, fabs(h32[m * 16 + n] - ref) / mag);
        worsts = fmax(worsts, fabs(hs[m * 16 + n] - ref) / mag);
      }
    hipFree(dA); hipFree(dB); hipFree(d32); hipFree(ds);
  }
  printf("max |err| / sum|terms|:  fp32 MFMA %.3e   bf16x3 split MFMA %.3e   (fp32 eps = 5.96e-08)\n", worst32, worsts);
  return 0;
}
