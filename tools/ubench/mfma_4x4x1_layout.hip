// Register layout and issue cost of v_mfma_f32_4x4x1_16b_f32 on gfx950 (the back-projection of amis_backward_mfma.hip rests on it):
// 16 blocks of a 4x1 by 1x4 outer product.  Prints, for every lane, which (A lane, B lane) product lands in each of its 4 result
// registers, then times a chain of dependent and of four interleaved accumulations.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_4x4x1_layout.hip -o /tmp/mfma4 && /tmp/mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((vector_size(16)));

__global__ void layout(float* out) {
  const int l = threadIdx.x;
  const float a = (float)(l + 1), b = 1000.f * (float)(l + 1);
  floatx4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = d[i];
}

template <int CHAINS>
__global__ void rate(float* out, int n) {
  const float a = (float)threadIdx.x * 1e-3f, b = 1.0001f;
  floatx4 d[CHAINS];
  for (int c = 0; c < CHAINS; ++c) d[c] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int k = 0; k < 12 / CHAINS; ++k)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) d[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += d[c][0] + d[c][1] + d[c][2] + d[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Per iteration: MF = 1: 12 v_mfma_f32_4x4x1 (4 chains), MF = 2: 12 v_mfma_f32_16x16x32_bf16 (4 chains); VA = 1: 36 independent
// v_pk_fma_f32, VA = 2: 72 independent v_fma_f32 (the same arithmetic); interleaved one MFMA : 3 packed / 6 scalar.
template <int MF, int VA>
__global__ void overlap(float* out, int n) {
  typedef float f32x2 __attribute__((vector_size(8)));
  typedef __bf16 bf16x8 __attribute__((vector_size(16)));
  typedef unsigned u32x4 __attribute__((vector_size(16)));
  const float a = (float)threadIdx.x * 1e-3f, b = 1.0001f;
  const u32x4 ua = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, ub = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  floatx4 d[4];
  f32x2 v[12];
  for (int c = 0; c < 4; ++c) d[c] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < 12; ++c) v[c] = f32x2{a + c, a - c};
  const f32x2 m = {b, 0.9999f}, k = {1e-6f, -1e-6f};
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      if (MF == 1) d[q & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d[q & 3], 0, 0, 0);
      if (MF == 2) d[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), d[q & 3], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        f32x2& x = v[(3 * q + e) % 12];
        if (VA == 1) x = __builtin_elementwise_fma(x, m, k);
        if (VA == 2) { x[0] = __builtin_fmaf(x[0], m[0], k[0]); x[1] = __builtin_fmaf(x[1], m[1], k[1]); }
      }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += d[c][0] + d[c][1] + d[c][2] + d[c][3];
  for (int c = 0; c < 12; ++c) s += v[c][0] + v[c][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MF, int VA>
static float time_overlap(float* o, int n, int waves, hipEvent_t e0, hipEvent_t e1) {
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    overlap<MF, VA><<<256, 256 * waves>>>(o, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1e6f / ((float)n * (float)waves);
}

int main() {
  float* o;
  hipMalloc(&o, sizeof(float) * 1024 * 1024);
  layout<<<1, 64>>>(o);
  float h[256];
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int al = 4 * (l >> 2) + i, bl = l;      // expected: D[i] of lane l = A(lane 4 * block + i) * B(lane l)
      const float want = (float)(al + 1) * 1000.f * (float)(bl + 1);
      if (h[l * 4 + i] != want) ok = 0;
    }
  for (int l = 0; l < 8; ++l) printf("lane %2d: %10.0f %10.0f %10.0f %10.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  printf("{\"layout_D_i_of_lane_l_is_A_lane_4blk_plus_i_times_B_lane_l\": %s}\n", ok ? "true" : "false");
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 20000;
  for (int waves = 1; waves <= 4; waves *= 2) {
    float ms[3];
    for (int v = 0; v < 3; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (v == 0) rate<1><<<256, 256 * waves>>>(o, n);
        if (v == 1) rate<2><<<256, 256 * waves>>>(o, n);
        if (v == 2) rate<4><<<256, 256 * waves>>>(o, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[v], e0, e1);
      }
    }
    // one workgroup per CU, `waves` waves per SIMD; 12 n MFMAs per wave
    printf("{\"waves_per_simd\": %d, \"ns_per_mfma_per_simd\": {\"1_chain\": %.2f, \"2_chains\": %.2f, \"4_chains\": %.2f}}\n", waves,
           ms[0] * 1e6 / (12.0 * n * waves), ms[1] * 1e6 / (12.0 * n * waves), ms[2] * 1e6 / (12.0 * n * waves));
  }
  // does a wave's (or its neighbour's) arithmetic run underneath these MFMAs?  ns per iteration and wave, both together against each alone
  for (int waves = 1; waves <= 2; ++waves) {
    printf("{\"waves_per_simd\": %d, \"alone\": {\"12_mfma_4x4x1\": %.1f, \"12_mfma_bf16_16x16x32\": %.1f, \"36_pk_fma\": %.1f, \"72_fma\": %.1f},\n", waves,
           time_overlap<1, 0>(o, n, waves, e0, e1), time_overlap<2, 0>(o, n, waves, e0, e1), time_overlap<0, 1>(o, n, waves, e0, e1),
           time_overlap<0, 2>(o, n, waves, e0, e1));
    printf(" \"together\": {\"4x4x1+pk\": %.1f, \"4x4x1+fma\": %.1f, \"bf16+pk\": %.1f, \"bf16+fma\": %.1f}}\n",
           time_overlap<1, 1>(o, n, waves, e0, e1), time_overlap<1, 2>(o, n, waves, e0, e1), time_overlap<2, 1>(o, n, waves, e0, e1),
           time_overlap<2, 2>(o, n, waves, e0, e1));
  }
  return ok ? 0 : 1;
}
