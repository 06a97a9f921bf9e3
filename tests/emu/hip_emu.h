// tests/emu/hip_emu.h -- TEST INFRASTRUCTURE ONLY (never part of the product build).
//
// A tiny single-OS-thread emulation of the HIP execution model, used to run the kernel sources of
// epro-pnp_amd/csrc on the CPU of the build container (which has no GPU) so that kernel LOGIC can be
// debugged and regression-tested under `pytest -m "not gpu"`.  Every GPU thread of a workgroup is a
// ucontext fiber; __syncthreads() and the wave-level exchange primitives are cooperative yields, so
// execution is deterministic and race-free.  Workgroups run one after another.
//
// The product library (libepropnp_hip.so) is compiled by hipcc from the same sources WITHOUT this header;
// the Python package never loads the emulation build (tests load it explicitly through ctypes).
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define EPROPNP_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
// host-side runtime calls the launchers make: the emulated "device" is this process (one compute unit: workgroups run one
// after another), stream-ordered allocations are plain malloc / free
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
enum { hipDeviceAttributeMultiprocessorCount = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 2 };
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 1; return 0; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { *p = std::malloc(n ? n : 1); return *p ? 0 : 1; }
inline hipError_t hipFreeAsync(void* p, hipStream_t) { std::free(p); return 0; }
// host-mapped memory of the default status word: plain host memory here; graph-capture modes do not exist
enum { hipHostMallocMapped = 1, hipHostMallocCoherent = 2 };
typedef int hipStreamCaptureMode;
enum { hipStreamCaptureModeRelaxed = 2 };
inline hipError_t hipThreadExchangeStreamCaptureMode(hipStreamCaptureMode*) { return 0; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? 0 : 1; }
inline hipError_t hipHostFree(void* p) { std::free(p); return 0; }
// events (per-stage timing inside the library): recorded, never timed -- every stage reads 0 ms
typedef int hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

namespace emu {

struct Idx3 {
  unsigned x, y, z;
};

struct Fiber {
  ucontext_t ctx;
  bool done;
  Idx3 tid;
};

struct Block {
  std::vector<Fiber> fibers;
  std::vector<char> stacks;
  int cur = 0, alive = 0, bar_count = 0, bar_gen = 0;
  std::vector<int> w_alive, w_count, w_gen;
  std::vector<std::array<uint64_t, 64>> w_buf;
  Idx3 bid{0, 0, 0};
  dim3 bdim, gdim;
  ucontext_t sched;
  std::function<void()> body;
  std::vector<char> dyn_smem;
};

inline Block& blk() {
  static Block b;
  return b;
}

constexpr size_t kStack = 512 * 1024;

inline void yield() {
  Block& b = blk();
  swapcontext(&b.fibers[b.cur].ctx, &b.sched);
}

inline void trampoline() {
  Block& b = blk();
  b.body();
  Fiber& f = b.fibers[b.cur];
  f.done = true;
  b.alive--;
  b.w_alive[f.tid.x / 64]--;
  swapcontext(&f.ctx, &b.sched);
}

inline void syncthreads() {
  Block& b = blk();
  int gen = b.bar_gen;
  b.bar_count++;
  while (b.bar_gen == gen) {
    if (b.bar_count >= b.alive) {
      b.bar_count = 0;
      b.bar_gen++;
      break;
    }
    yield();
  }
}

inline void wave_sync() {
  Block& b = blk();
  int w = b.fibers[b.cur].tid.x / 64;
  int gen = b.w_gen[w];
  b.w_count[w]++;
  while (b.w_gen[w] == gen) {
    if (b.w_count[w] >= b.w_alive[w]) {
      b.w_count[w] = 0;
      b.w_gen[w]++;
      break;
    }
    yield();
  }
}

// fibers of a block run one at a time on one thread: a plain read-modify-write is atomic
inline float atomic_add(float* p, float v) { float o = *p; *p = o + v; return o; }

template <class T>
inline T shfl(T v, int src) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  Block& b = blk();
  unsigned t = b.fibers[b.cur].tid.x;
  int w = t / 64, lane = t % 64;
  uint64_t bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  b.w_buf[w][lane] = bits;
  wave_sync();
  T r;
  uint64_t got = b.w_buf[w][src & 63];
  std::memcpy(&r, &got, sizeof(T));
  wave_sync();
  return r;
}

inline void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> body) {
  Block& b = blk();
  unsigned nthreads = block.x * block.y * block.z;
  if (block.y != 1 || block.z != 1 || nthreads % 64 != 0) {
    std::fprintf(stderr, "emu: only 1-D blocks that are multiples of 64 are supported\n");
    std::abort();
  }
  unsigned nw = nthreads / 64;
  b.bdim = block;
  b.gdim = grid;
  b.body = std::move(body);
  b.dyn_smem.assign(smem + 64, 0);
  if (b.stacks.size() < nthreads * kStack) b.stacks.resize(nthreads * kStack);
  b.fibers.resize(nthreads);
  for (unsigned gz = 0; gz < grid.z; ++gz)
    for (unsigned gy = 0; gy < grid.y; ++gy)
      for (unsigned gx = 0; gx < grid.x; ++gx) {
        b.bid = Idx3{gx, gy, gz};
        b.alive = (int)nthreads;
        b.bar_count = 0;
        b.w_alive.assign(nw, 64);
        b.w_count.assign(nw, 0);
        b.w_gen.assign(nw, 0);
        b.w_buf.assign(nw, std::array<uint64_t, 64>{});
        for (unsigned t = 0; t < nthreads; ++t) {
          Fiber& f = b.fibers[t];
          f.done = false;
          f.tid = Idx3{t, 0, 0};
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = b.stacks.data() + (size_t)t * kStack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, (void (*)())trampoline, 0);
        }
        while (b.alive > 0) {
          for (unsigned t = 0; t < nthreads; ++t) {
            if (b.fibers[t].done) continue;
            b.cur = (int)t;
            swapcontext(&b.sched, &b.fibers[t].ctx);
          }
        }
      }
}

}  // namespace emu

#define threadIdx (emu::blk().fibers[emu::blk().cur].tid)
#define blockIdx (emu::blk().bid)
#define blockDim (emu::blk().bdim)
#define gridDim (emu::blk().gdim)
inline void __syncthreads() { emu::syncthreads(); }
inline float atomicAdd(float* p, float v) { return emu::atomic_add(p, v); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
inline int atomicExch(int* p, int v) { const int o = *p; *p = v; return o; }
inline void __threadfence() {}
inline int atomicOr(int* p, int v) { const int o = *p; *p = o | v; return o; }        // one fiber at a time: plain RMW
inline int atomicMin(int* p, int v) { const int o = *p; *p = v < o ? v : o; return o; }

#define PNP_LAUNCH(kernel, grid, block, smem, stream, ...) \
  emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define PNP_DYN_SMEM(type, name) \
  type* name = reinterpret_cast<type*>((reinterpret_cast<uintptr_t>(emu::blk().dyn_smem.data()) + 15) & ~uintptr_t(15))

using std::max;
using std::min;
struct float2 {
  float x, y;
};
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct float4 {
  float x, y, z, w;
};
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

// v_mfma_f32_16x16x4_f32: D(16x16) = A(16x4) B(4x16) + C.  Lane l holds A[l&15][l>>4], B[l>>4][l&15] and
// D[4*(l>>4)+r][l&15], r = 0..3.  k-ordered fmaf chain, like the hardware.
typedef float floatx4_emu __attribute__((vector_size(16)));
typedef unsigned u32x4_emu __attribute__((vector_size(16)));
namespace emu {
inline floatx4_emu mfma_16x16x4(float a, float b, floatx4_emu c) {
  Block& blk_ = blk();
  unsigned t = blk_.fibers[blk_.cur].tid.x;
  int w = t / 64, lane = t % 64;
  float av[4][4], bv[4];
  uint64_t bits = 0;
  std::memcpy(&bits, &a, 4);
  blk_.w_buf[w][lane] = bits;
  wave_sync();
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 4; ++k) {
      uint64_t g = blk_.w_buf[w][k * 16 + 4 * (lane >> 4) + r];
      std::memcpy(&av[r][k], &g, 4);
    }
  wave_sync();
  bits = 0;
  std::memcpy(&bits, &b, 4);
  blk_.w_buf[w][lane] = bits;
  wave_sync();
  for (int k = 0; k < 4; ++k) {
    uint64_t g = blk_.w_buf[w][k * 16 + (lane & 15)];
    std::memcpy(&bv[k], &g, 4);
  }
  wave_sync();
  floatx4_emu d = c;
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 4; ++k) d[r] = fmaf(av[r][k], bv[k], d[r]);
  return d;
}

// v_mfma_f32_16x16x32_bf16: D(16x16) = A(16x32) B(32x16) + C.  Lane l holds the 8 bf16 A[l&15][8*(l>>4) .. +7] and
// B[8*(l>>4) .. +7][l&15] as 4 dwords each (element 2i = low half of dword i); D layout as mfma_16x16x4.  Products of bf16
// pairs are exact in fp32; the 32-term sum is accumulated in double and rounded once (the hardware's order is not specified).
inline floatx4_emu mfma_16x16x32_bf16(u32x4_emu a, u32x4_emu b, floatx4_emu c) {
  Block& blk_ = blk();
  unsigned t = blk_.fibers[blk_.cur].tid.x;
  int w = t / 64, lane = t % 64;
  unsigned av[64][4], bv[64][4];
  for (int half = 0; half < 2; ++half) {       // two dwords per exchange round (w_buf holds 64 bits per lane)
    blk_.w_buf[w][lane] = (uint64_t)a[2 * half] | ((uint64_t)a[2 * half + 1] << 32);
    wave_sync();
    for (int l = 0; l < 64; ++l) { av[l][2 * half] = (unsigned)blk_.w_buf[w][l]; av[l][2 * half + 1] = (unsigned)(blk_.w_buf[w][l] >> 32); }
    wave_sync();
    blk_.w_buf[w][lane] = (uint64_t)b[2 * half] | ((uint64_t)b[2 * half + 1] << 32);
    wave_sync();
    for (int l = 0; l < 64; ++l) { bv[l][2 * half] = (unsigned)blk_.w_buf[w][l]; bv[l][2 * half + 1] = (unsigned)(blk_.w_buf[w][l] >> 32); }
    wave_sync();
  }
  auto elem = [](const unsigned (&v)[4], int e) {
    const unsigned bits = ((e & 1) ? (v[e >> 1] >> 16) : (v[e >> 1] & 0xffffu)) << 16;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
  };
  floatx4_emu d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (lane >> 4) + r, col = lane & 15;
    double acc = d[r];
    for (int g = 0; g < 4; ++g)
      for (int e = 0; e < 8; ++e) acc += (double)elem(av[g * 16 + row], e) * (double)elem(bv[g * 16 + col], e);
    d[r] = (float)acc;
  }
  return d;
}
}  // namespace emu

// ---- the gfx950 compiler builtins the kernel headers use, under their own names ------------------------------------------
// The product headers (wave_ops.h, pnp_math.h, amis_common.h, pnp_sweep.h) are written against the AMDGPU builtins only and hold
// no test branches; this section gives g++ a host function of the same name and meaning for each of them.  Cross-lane builtins
// are cooperative exchanges between the fibers of a wave; the 1-ulp hardware approximations become the libm functions.
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <class T> inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T> inline void __hip_atomic_store(T* p, T v, int, int) { *p = v; }
template <class T> inline T __hip_atomic_fetch_or(T* p, T v, int, int) { const T o = *p; *p = o | v; return o; }
template <class T> inline T __hip_atomic_fetch_min(T* p, T v, int, int) { const T o = *p; *p = v < o ? v : o; return o; }
template <class T> inline T __builtin_nontemporal_load(const T* p) { return *p; }

inline int __float_as_int(float x) { int i; std::memcpy(&i, &x, 4); return i; }
inline float __int_as_float(int i) { float x; std::memcpy(&x, &i, 4); return x; }

inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
inline float __builtin_amdgcn_logf(float x) { return log2f(x); }                  // v_log_f32 is base 2
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __builtin_amdgcn_sinf(float rev) { return (float)sin(6.283185307179586 * (double)rev); }      // argument in revolutions
inline float __builtin_amdgcn_cosf(float rev) { return (float)cos(6.283185307179586 * (double)rev); }
// v_med3_f32: the median; with a NaN among the operands the minimum of the others (fminf ignores a NaN)
inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {
  if (a != a || b != b || c != c) return fminf(fminf(a, b), c);
  return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}
template <class V> inline V __builtin_elementwise_fma(V a, V b, V c) {
  V r;
  for (unsigned i = 0; i < sizeof(V) / sizeof(float); ++i) r[i] = fmaf(a[i], b[i], c[i]);
  return r;
}
// v_perm_b32: byte k of the result is byte sel[k] of the 8 bytes {src0 (bytes 7..4), src1 (bytes 3..0)}; 0x0c selects 0x00
inline unsigned __builtin_amdgcn_perm(unsigned src0, unsigned src1, unsigned sel) {
  const uint64_t both = ((uint64_t)src0 << 32) | src1;
  unsigned r = 0;
  for (int k = 0; k < 4; ++k) {
    const unsigned s = (sel >> (8 * k)) & 0xffu;
    const unsigned byte = s < 8 ? (unsigned)(both >> (8 * s)) & 0xffu : (s == 0x0c ? 0u : 0xffu);
    r |= byte << (8 * k);
  }
  return r;
}

inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_sleep(int) {}
inline void __builtin_amdgcn_fence(int, const char*) {}
inline void __builtin_amdgcn_wave_barrier() { emu::wave_sync(); }      // the fibers of a wave do not run in lock step: meet here
// Workgroups run one after another, so a word another workgroup has not written yet never arrives: the clock jumps, a poll ends.
inline unsigned long long __builtin_amdgcn_s_memtime() {
  static unsigned long long t = 0;
  return t += 1ull << 40;
}
// wave-uniform by contract of its callers
inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
inline int __builtin_amdgcn_readlane(int x, int src) { return emu::shfl(x, src); }
// Callers sit in divergent code and only test `ballot(p) == 0` to leave a retry loop early; a fiber cannot wait for lanes that
// took another branch, so it sees its own bit only (it leaves when IT is done, which changes no result).
inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) {
  return p ? 1ull << (emu::blk().fibers[emu::blk().cur].tid.x & 63u) : 0ull;
}
// DPP controls in use: quad_perm (0x00..0xff) and row_ror:n (0x121..0x12f); all rows / banks enabled
inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool) {
  const int l = (int)(emu::blk().fibers[emu::blk().cur].tid.x & 63u);
  int from;
  if (ctrl >= 0 && ctrl <= 0xff) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
  else if (ctrl >= 0x121 && ctrl <= 0x12f) from = (l & ~15) | ((l - (ctrl & 15)) & 15);
  else { std::fprintf(stderr, "emu: DPP control 0x%x is not emulated\n", ctrl); std::abort(); }
  return emu::shfl(src, from);
}
inline floatx4_emu __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, floatx4_emu c, int, int, int) { return emu::mfma_16x16x4(a, b, c); }
typedef unsigned short __bf16;           // storage only (g++ 11 has no such type); the emulation reads the bits
typedef __bf16 bf16x8_emu __attribute__((vector_size(16)));
inline floatx4_emu __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf16x8_emu a, bf16x8_emu b, floatx4_emu c, int, int, int) {
  return emu::mfma_16x16x32_bf16(__builtin_bit_cast(u32x4_emu, a), __builtin_bit_cast(u32x4_emu, b), c);
}
