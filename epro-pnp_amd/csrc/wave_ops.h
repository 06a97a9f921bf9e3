// wave_ops.h -- wavefront (64 lanes) and workgroup reduction / broadcast primitives for gfx950.
//
// On the device these are DPP row rotations (v_add_f32 ... row_ror) plus v_readlane, i.e. they never touch
// LDS memory.  The reduction order is fixed, so results are run-to-run deterministic.
// Everything here is written against the compiler's AMDGPU builtins, once.
#pragma once

#include <hip/hip_runtime.h>
#ifndef PNP_LAUNCH      // how a kernel is launched and how it names its dynamic LDS: a build may bring its own pair
#define PNP_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)
#define PNP_DYN_SMEM(type, name)                                                  \
  extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_smem[]; \
  type* name = reinterpret_cast<type*>(name##_raw_smem)
#endif

namespace pnp {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// The instruction scheduler may not move anything across this point.  Between the per-point bodies of a sweep over
// register-resident points it keeps point k's arithmetic ahead of point k+1's, so that the s_waitcnt in front of it waits for
// point k's loads only and the arithmetic runs underneath the loads still in flight (otherwise the scheduler interleaves the
// independent per-point chains and every wave waits for ALL of its loads before its first FMA).
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// wave index as a wave-uniform (scalar-register) value: lets the compiler keep everything derived from it in SGPRs
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

// value of `x` held by lane `src` (wave-uniform `src`) -> scalar register
__device__ __forceinline__ float wave_bcast(float x, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}
__device__ __forceinline__ int wave_bcast(int x, int src) { return __builtin_amdgcn_readlane(x, src); }

// sum over the 64 lanes, result in every lane.  row_ror:8/4/2/1 inside each 16-lane row, then 4 readlanes.
__device__ __forceinline__ float wave_sum(float x) {
  x += dpp_mov<0x128>(x);
  x += dpp_mov<0x124>(x);
  x += dpp_mov<0x122>(x);
  x += dpp_mov<0x121>(x);
  return (wave_bcast(x, 0) + wave_bcast(x, 16)) + (wave_bcast(x, 32) + wave_bcast(x, 48));
}

__device__ __forceinline__ float wave_max(float x) {
  x = fmaxf(x, dpp_mov<0x128>(x));
  x = fmaxf(x, dpp_mov<0x124>(x));
  x = fmaxf(x, dpp_mov<0x122>(x));
  x = fmaxf(x, dpp_mov<0x121>(x));
  return fmaxf(fmaxf(wave_bcast(x, 0), wave_bcast(x, 16)), fmaxf(wave_bcast(x, 32), wave_bcast(x, 48)));
}

// sum over the 16 lanes of a DPP row (every lane of the row receives it)
__device__ __forceinline__ float row_sum16(float x) {
  x += dpp_mov<0x128>(x);
  x += dpp_mov<0x124>(x);
  x += dpp_mov<0x122>(x);
  x += dpp_mov<0x121>(x);
  return x;
}

// Row sums of FOUR values at once: lane l of a 16-lane row receives the row total of a[l & 3].  The first two butterfly steps
// halve the values instead of repeating them -- each lane keeps the two (then the one) it will own and hands the others to its
// partner -- so the four sums cost 5 DPP adds (half-rate instructions on gfx950) + 6 selects instead of 16 DPP adds.
__device__ __forceinline__ float row_sum16_of4(const float (&a)[4]) {
  const int l = lane_id();
  const bool o1 = (l & 1) != 0, o2 = (l & 2) != 0;
  float k01 = o1 ? a[1] : a[0], s01 = o1 ? a[0] : a[1];
  float k23 = o1 ? a[3] : a[2], s23 = o1 ? a[2] : a[3];
  k01 += dpp_mov<0xB1>(s01);      // quad_perm:[1,0,3,2]
  k23 += dpp_mov<0xB1>(s23);
  float k = o2 ? k23 : k01;
  const float s = o2 ? k01 : k23;
  k += dpp_mov<0x4E>(s);          // quad_perm:[2,3,0,1]
  k += dpp_mov<0x128>(k);         // row_ror:8
  k += dpp_mov<0x124>(k);         // row_ror:4
  return k;
}

// sum over the 4 lanes of a quad (lanes 4q .. 4q+3; every lane of the quad receives (x0 + x1) + (x2 + x3))
__device__ __forceinline__ float quad_sum(float x) {
  x += dpp_mov<0xB1>(x);      // quad_perm:[1,0,3,2]
  x += dpp_mov<0x4E>(x);      // quad_perm:[2,3,0,1]
  return x;
}

// max over the 16 lanes of a DPP row (every lane of the row receives it)
__device__ __forceinline__ float row_max16(float x) {
  x = fmaxf(x, dpp_mov<0x128>(x));
  x = fmaxf(x, dpp_mov<0x124>(x));
  x = fmaxf(x, dpp_mov<0x122>(x));
  x = fmaxf(x, dpp_mov<0x121>(x));
  return x;
}

// min over the 16 lanes of a DPP row / over the wave, unsigned (packed sort keys)
__device__ __forceinline__ unsigned row_min16_u32(unsigned x) {
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xf, 0xf, false));
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, false));
  return x;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
  x = row_min16_u32(x);
  const unsigned a = (unsigned)wave_bcast((int)x, 0), b = (unsigned)wave_bcast((int)x, 16);
  const unsigned c = (unsigned)wave_bcast((int)x, 32), d = (unsigned)wave_bcast((int)x, 48);
  const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
}

// ---- matrix core and packed-pair primitives (the AMIS sweeps) ------------------------------------------------------
// v_mfma_f32_16x16x4_f32: D(16x16) = A(16x4) B(4x16) + C, exact f32 (a k-ordered fmaf chain).  Lane l holds A[l&15][l>>4],
// B[l>>4][l&15] and D[4*(l>>4)+r][l&15], r = 0..3.
typedef float floatx4 __attribute__((vector_size(16)));
typedef float f32x2 __attribute__((vector_size(8)));
__device__ __forceinline__ floatx4 mfma_16x16x4(float a, float b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// two point-poses at a time on 2-vectors, so that the multiplies / FMAs become v_pk_mul_f32 / v_pk_fma_f32
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// exact-zero test of a wave-uniform float without a vector compare
__device__ __forceinline__ bool uniform_is_zero(float a) { return __builtin_amdgcn_readfirstlane(__float_as_int(a)) == 0; }

// ---- bf16x3-split projection: v_mfma_f32_16x16x32_bf16 --------------------------------------------------------------------
// An fp32 product sum over k = 0..3 as ONE 32-deep bf16 MFMA: each fp32 operand is the sum of three bf16 pieces (split by
// TRUNCATION, so the pieces sum to the value exactly), and the 8 slots of a real k carry the products
//   a1b1 a2b1 | a1b2 a2b2 | a1b3 a2b3 | a3b1 a3b2      (a3b3 ~ 2^-32 of the term is dropped)
// -- fp32-level accuracy (tools/ubench/bf16_split_mfma.hip: 1.4e-7 against 1.2e-7 relative for the fp32 MFMA) at 55 % of the
// fp32 MFMA's time (4 passes instead of 8).  Lane l holds slots 8 (l >> 4) .. +7 of row / column l & 15 as 4 dwords (element 2i =
// low half of dword i); D as mfma_16x16x4.  The operands are plain vector VALUES and the MFMA is the compiler's builtin: its
// hazard recogniser keeps the distances this instruction needs on gfx950 (8 wait states before a VALU read of a result, 1
// behind a VALU-written source: the ISA of a probe shows `s_nop 7` / `s_nop 0`, the distances tools/ubench/mfma_bf16_hazard.hip
// measured).  Do NOT route an operand or a result through an inline asm: an asm that takes a result in-out hides the MFMA ->
// VALU dependence from the recogniser (the asm "redefines" the register), which is what made the round-3 attempts at this
// projection wrong and run-to-run different (profiles/r04_bwd_bf16_projection.txt).
typedef unsigned u32x4 __attribute__((vector_size(16)));
__device__ __forceinline__ floatx4 mfma_16x16x32_bf16(u32x4 a, u32x4 b, floatx4 c) {
  typedef __bf16 bf16x8_ __attribute__((vector_size(16)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}
// {hi16(lo_src), hi16(hi_src)} as one v_perm_b32
__device__ __forceinline__ unsigned bf16_pack_hi(unsigned lo_src, unsigned hi_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u); }
__device__ __forceinline__ void bf16_split3(float x, unsigned& p1, unsigned& p2, unsigned& p3) {      // pieces in the HIGH halves
  unsigned u;
  __builtin_memcpy(&u, &x, 4);
  p1 = u & 0xffff0000u;
  float f1;
  __builtin_memcpy(&f1, &p1, 4);
  const float r1 = x - f1;                       // exact: the low 16 mantissa bits
  __builtin_memcpy(&u, &r1, 4);
  p2 = u & 0xffff0000u;
  float f2;
  __builtin_memcpy(&f2, &p2, 4);
  const float r2 = r1 - f2;                      // exact, <= 8 significant bits left: a bf16 value
  __builtin_memcpy(&p3, &r2, 4);
}
__device__ __forceinline__ u32x4 bf16_split_a(float a) {      // (a1,a2) (a1,a2) (a1,a2) (a3,a3)
  unsigned a1, a2, a3;
  bf16_split3(a, a1, a2, a3);
  const unsigned w0 = bf16_pack_hi(a1, a2);
  return u32x4{w0, w0, w0, bf16_pack_hi(a3, a3)};
}
__device__ __forceinline__ u32x4 bf16_split_b(float b) {      // (b1,b1) (b2,b2) (b3,b3) (b1,b2)
  unsigned b1, b2, b3;
  bf16_split3(b, b1, b2, b3);
  return u32x4{bf16_pack_hi(b1, b1), bf16_pack_hi(b2, b2), bf16_pack_hi(b3, b3), bf16_pack_hi(b1, b2)};
}
// operand type / split / product of the two projection flavours, so that a kernel is written once
template <bool BF16> struct ProjOp;
template <> struct ProjOp<false> {
  typedef float T;
  static __device__ __forceinline__ T a(float x) { return x; }
  static __device__ __forceinline__ T b(float x) { return x; }
  static __device__ __forceinline__ floatx4 mma(T x, T y, floatx4 c) { return mfma_16x16x4(x, y, c); }
};
template <> struct ProjOp<true> {
  typedef u32x4 T;
  static __device__ __forceinline__ T a(float x) { return bf16_split_a(x); }
  static __device__ __forceinline__ T b(float x) { return bf16_split_b(x); }
  static __device__ __forceinline__ floatx4 mma(T x, T y, floatx4 c) { return mfma_16x16x32_bf16(x, y, c); }
};

// ---- words exchanged between the workgroups of ONE launch (the split-over-workgroups variants of the AMIS forward and
// the LM solve) ------------------------------------------------------------------------------------------------------
// A slot is pre-filled with kXwgEmpty by the launcher; its producer overwrites it with the payload (never kXwgEmpty: NaNs
// are stored canonical), so the data is its own arrival flag.  Every access is a relaxed AGENT-scope atomic (sc1 stores /
// loads: they bypass the per-CU L1 and the non-coherent L2 lines of other XCDs); no release / acquire fences -- at agent
// scope those write back / invalidate whole caches, 20 us per exchange (profiles/r03_fwd_split_timing.txt).
// A consumer waits for a slot for at most `timeout_cycles` shader cycles and then gets kXwgEmpty back: the caller computes
// the missing value ITSELF from the object's points (same lanes, same order: same bits), so correctness never depends on
// the sibling workgroups being resident -- a CU mask, a partitioned GPU or a foreign kernel holding CUs costs time, not
// results.  (CPU emulation: workgroups run one after another, a slot is either there or it is not.)
constexpr unsigned kXwgEmpty = 0xffffffffu;

__device__ __forceinline__ unsigned f32_bits(float x) { unsigned u; __builtin_memcpy(&u, &x, 4); return u; }
__device__ __forceinline__ float bits_f32(unsigned u) { float x; __builtin_memcpy(&x, &u, 4); return x; }
// payload of a float: its bits, NaNs canonical (so that no payload equals kXwgEmpty, a negative NaN with a full payload)
__device__ __forceinline__ unsigned xwg_payload(float x) { return (x != x) ? 0x7fc00000u : f32_bits(x); }

__device__ __forceinline__ void xwg_store(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned xwg_poll(const unsigned* p, unsigned timeout_cycles) {
  unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v != kXwgEmpty || timeout_cycles == 0) return v;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  do {
    __builtin_amdgcn_s_sleep(1);
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } while (v == kXwgEmpty && (unsigned long long)__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)timeout_cycles);
  return v;
}

// Orders this wave's earlier LDS writes before its later LDS reads of OTHER lanes' data.  The hardware executes a
// wave's DS instructions in order, so no s_barrier is needed; this only stops the compiler from reordering.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Sum NV per-thread values over the whole workgroup; every thread receives the totals.
// `scratch` must hold NV * (blockDim.x / 64) floats of LDS.  Two barriers (none for a single-wave group).
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* scratch) {
  const int nw = (int)(blockDim.x >> 6);
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  if (nw == 1) return;
  const int w = wave_id();
  if (lane_id() == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) scratch[w * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = scratch[i];
    for (int k = 1; k < nw; ++k) s += scratch[k * NV + i];
    v[i] = s;
  }
  __syncthreads();
}

// Same contract as block_sum, but the per-wave reduction goes through LDS "transposed": every lane parks its NV
// partials (one ds_write each), lane i < NV then adds up row i (16 x ds_read_b128 + 63 adds), and the NV totals are
// re-broadcast.  ~4x fewer VALU instructions than NV DPP/readlane chains (DPP adds and v_readlane issue at half the
// rate of an FMA on gfx950).  `lds` must hold nw * kSumTStride<NV> floats, 16-B aligned; needs NV <= 64.
constexpr int kSumTRow = 68;   // 64 lanes + 4 floats padding (keeps rows 16-B aligned)
template <int NV>
constexpr int kSumTStride = NV * kSumTRow + ((NV + 3) / 4) * 4;

template <int NV>
__device__ __forceinline__ void block_sum_t(float (&v)[NV], float* lds) {
  static_assert(NV <= 64, "one lane per reduced value");
  const int nw = (int)(blockDim.x >> 6), w = wave_id(), lane = lane_id();
  float* rows = lds + w * kSumTStride<NV>;
  float* tot = rows + NV * kSumTRow;
#pragma unroll
  for (int i = 0; i < NV; ++i) rows[i * kSumTRow + lane] = v[i];
  wave_lds_fence();
  if (lane < NV) {
    const float4* r = reinterpret_cast<const float4*>(rows + lane * kSumTRow);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 q = r[k];
      acc[k & 3] += (q.x + q.y) + (q.z + q.w);
    }
    tot[lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
  if (nw == 1) wave_lds_fence(); else __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float sum = lds[NV * kSumTRow + i];
    for (int k = 1; k < nw; ++k) sum += lds[k * kSumTStride<NV> + NV * kSumTRow + i];
    v[i] = sum;
  }
  if (nw == 1) wave_lds_fence(); else __syncthreads();
}

__device__ __forceinline__ float block_max(float x, float* scratch) {
  const int nw = (int)(blockDim.x >> 6);
  x = wave_max(x);
  if (nw == 1) return x;
  if (lane_id() == 0) scratch[wave_id()] = x;
  __syncthreads();
  float m = scratch[0];
  for (int k = 1; k < nw; ++k) m = fmaxf(m, scratch[k]);
  __syncthreads();
  return m;
}

}  // namespace pnp
