// Shared device/host helpers for libttk_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ttk.h"

namespace ttk {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU

void set_error(const char* fmt, ...);

// Host-side argument check: records the message and returns a negative code from the entry point.
#define TTK_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::ttk::set_error(__VA_ARGS__);    \
      return -1;                        \
    }                                   \
  } while (0)

// After a launch: report the HIP error of this thread, if any (no synchronisation).
#define TTK_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      ::ttk::set_error("%s: %s", name, hipGetErrorString(e_));              \
      return (int)e_;                                                       \
    }                                                                       \
    return 0;                                                               \
  } while (0)

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Environment switches.  The PRODUCT library reads exactly one variable, TTK_DETERMINISTIC (a documented mode: DESIGN.md 4.8).  Every
// A/B switch (kernel selection, forced tilings: TTK_GEMM, TTK_GEMM_R, TTK_R_RBLK, TTK_R_ALL, TTK_WGRAD_T, TTK_STEM, TTK_STEM7_VALU,
// TTK_FUSED_FP32, TTK_CONV_WIDE64, TTK_DW_COLTILE) exists only in experiment builds (-DTTK_EXPERIMENTS: tools/exp/build_variants.sh);
// in the product exp_env() is a constant and the branches behind it fold away.
#if defined(TTK_EXPERIMENTS)
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif
inline bool deterministic_mode() {
  static const bool det = [] { const char* e = getenv("TTK_DETERMINISTIC"); return e && e[0] != '0'; }();
  return det;
}

// Activation layout of the MobileNet path (include/ttk.h, "channel blocks"): a tensor of M pixels x C channels is stored as
// [C / 32][M][32] - element (m, c) at ((c >> 5) * M + m) * 32 + (c & 31).  A depthwise workgroup's 32-channel slab and a GEMM's
// k32 step are then CONTIGUOUS runs (pixels x 128 B) instead of 128-byte pieces of 4 C-byte rows: the same kernels stream 10-20 %
// faster (profiles/r03_stream_sweep.txt: 5:1 read:write mix 4.9 TB/s in 128-byte pieces, 5.9-6.2 TB/s linear).  C = 32: plain [M][32].
constexpr int kCB = 32;
__host__ __device__ __forceinline__ size_t act_off(int64_t m, int c, int64_t M) { return ((size_t)(c >> 5) * (size_t)M + (size_t)m) * kCB + (c & (kCB - 1)); }
__host__ __device__ __forceinline__ size_t act_block_stride(int64_t M) { return (size_t)M * kCB; }  // elements between channel blocks

inline int elementwise_grid(int64_t items) {
  int64_t g = ceil_div(items, kBlock);
  if (g > TTK_MAX_PARTIAL_ROWS_ELEMENTWISE) g = TTK_MAX_PARTIAL_ROWS_ELEMENTWISE;
  if (g < 1) g = 1;
  return (int)g;
}

// Power-of-two scale S of an fp16-split GEMM operand (pwconv_f16.hip): bound * S lies in [2^14, 2^15), so every scaled
// value is below the fp16 maximum; 1 when the bound is unknown (<= 0, inf, nan).  |log2 S| <= 60: products of two scales
// and their reciprocals stay inside the fp32 exponent range.
__host__ __device__ inline float pow2_scale(float bound) {
  if (!(bound > 0.f) || bound > 3.0e38f) return 1.f;
  union { float f; unsigned u; } b;
  b.f = bound;
  int s = 14 - ((int)((b.u >> 23) & 0xffu) - 127);
  s = s > 60 ? 60 : (s < -60 ? -60 : s);
  b.u = (unsigned)(s + 127) << 23;
  return b.f;
}

// Raises the non-negative float at *slot (ordered like its bit pattern; zeroed once per step) to the maximum of v over
// the wavefront.  The read keeps the atomic off the common path (the slot is almost always already larger).
__device__ __forceinline__ void wave_raise_max(float* slot, float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  unsigned* u = reinterpret_cast<unsigned*>(slot);
  if ((threadIdx.x & 63) == 0 && __float_as_uint(v) > __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(u, __float_as_uint(v));
}

// Deterministic reductions: producers store their partial results row by row (partial[r][n]) instead of adding them
// atomically, and this kernel folds the rows in a fixed order:  out[i] (+)= partial[0][i] + partial[1][i] + ...
template <int kDummy = 0>
__global__ void __launch_bounds__(256) fold_partials_k(const float* __restrict__ partial, int rows, int64_t n, float* __restrict__ out,
                                                        int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = accumulate ? out[i] : 0.f;
  for (int r = 0; r < rows; ++r) a += partial[(size_t)r * n + i];
  out[i] = a;
}
// many rows, few outputs (the stem's 1024 workgroup rows of 3136 weights): 8 row groups per output, each summed in row order,
// then the 8 group sums in group order - still a fixed order, 8 times the parallelism and 32-float coalesced row reads
template <int kDummy = 0>
__global__ void __launch_bounds__(256) fold_partials_wide_k(const float* __restrict__ partial, int rows, int64_t n, float* __restrict__ out,
                                                             int accumulate) {
  __shared__ float sm[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + tx;
  float a = 0.f;
  if (i < n)
    for (int r = ty; r < rows; r += 8) a += partial[(size_t)r * n + i];
  sm[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && i < n) {
    float t = accumulate ? out[i] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sm[q][tx];
    out[i] = t;
  }
}
inline void launch_fold_partials(const float* partial, int rows, int64_t n, float* out, int accumulate, hipStream_t st) {
  if (rows >= 64 && n <= 65536)
    hipLaunchKernelGGL(fold_partials_wide_k<0>, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, partial, rows, n, out, accumulate);
  else
    hipLaunchKernelGGL(fold_partials_k<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, rows, n, out, accumulate);
}

// ---- float4 helpers -------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (non-temporal) 16-byte load: for tensors a kernel reads exactly once (tools/exp/slab_probe.hip: the
// depthwise tile pattern reads 6.2-6.6 TB/s this way against 5.5-5.9 TB/s with plain loads)
typedef float ttk_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt(const float* p) {
  const ttk_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ttk_f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 relu4(float4 a) {
  return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
}
// g * [a > 0]
__device__ __forceinline__ float4 mask4(float4 g, float4 a) {
  return make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f, a.z > 0.f ? g.z : 0.f, a.w > 0.f ? g.w : 0.f);
}

__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}

// ---- storage type of the activation tensors that cross HBM ------------------------------------------------------
// float (the reference's precision) for the fp32 kernel family; Act<bf16_t> (values rounded to nearest-even bf16 on their way out,
// widened on load) is what remains of rounds 2-5's bf16 STORAGE variant: only the stem of the bf16-compute path instantiates it
// (TTK_ACT_DISPATCH_STEM below).  Offsets are in elements either way.
typedef uint16_t bf16_t;  // storage only
typedef __bf16 ttk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ttk_f32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct Act;
template <> struct Act<float> {
  static constexpr bool kBf16 = false;
  __device__ __forceinline__ static float4 ld(const float* p) { return ld4(p); }
  __device__ __forceinline__ static float4 ldnt(const float* p) { return ld4nt(p); }
  __device__ __forceinline__ static float4 round(float4 v) { return v; }
  // (streaming / non-temporal stores of the outputs were measured: the depthwise forward, which re-reads what the previous kernel
  // left in L2 / MALL, went 0.72 -> 0.85 ms per step; the step 8.05 -> 8.13 ms)
  __device__ __forceinline__ static void st(float* p, float4 v) { st4(p, v); }
  __device__ __forceinline__ static float st1(float* p, float v) { *p = v; return v; }  // scalar store; returns the stored value
};
template <> struct Act<bf16_t> {
  static constexpr bool kBf16 = true;
  __device__ __forceinline__ static float4 widen(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  }
  __device__ __forceinline__ static float4 ld(const bf16_t* p) { return widen(*reinterpret_cast<const uint2*>(p)); }
  __device__ __forceinline__ static float4 ldnt(const bf16_t* p) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 u = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    return widen(make_uint2(u.x, u.y));
  }
  __device__ __forceinline__ static uint2 pack(float4 v) {  // round to nearest even (v_cvt_pk_bf16_f32)
    const ttk_bf16x2 a = __builtin_convertvector(ttk_f32x2{v.x, v.y}, ttk_bf16x2), b = __builtin_convertvector(ttk_f32x2{v.z, v.w}, ttk_bf16x2);
    return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
  }
  __device__ __forceinline__ static float4 round(float4 v) { return widen(pack(v)); }  // the value the consumer will read
  __device__ __forceinline__ static void st(bf16_t* p, float4 v) { *reinterpret_cast<uint2*>(p) = pack(v); }
  __device__ __forceinline__ static float st1(bf16_t* p, float v) {
    const __bf16 b = (__bf16)v;
    const uint16_t u = __builtin_bit_cast(uint16_t, b);
    *p = u;
    return __uint_as_float((unsigned)u << 16);
  }
};
// The storage flag of the C-ABI (TTK_STORE_* in ttk.h).  Round 6 retired the bf16 STORAGE variants of the fp32 kernels (`--precision bf16 | bf16-all`:
// slower than fp32, superseded by the bf16-compute path's own kernels, csrc/bc_*.hip): the depthwise / pointwise / pooling entry points take fp32
// tensors only and refuse the bits; the stem pair, which also serves the bf16-compute path (C = 32: the same bytes in either layout), takes fp32 or
// BOTH bits (activations and gradients bfloat16).
#define TTK_ACT_DISPATCH(flag, ...)                                                                                                             \
  do {                                                                                                                                          \
    TTK_REQUIRE(((flag) & 3) == 0, "bf16 activation storage under the fp32 kernels was retired (round 6): use the bf16-compute path (ttk_bc_*)"); \
    using ActT [[maybe_unused]] = float; using GradT [[maybe_unused]] = float; __VA_ARGS__;                                         \
  } while (0)
#define TTK_ACT_DISPATCH_STEM(flag, ...)                                                                                                        \
  do {                                                                                                                                          \
    TTK_REQUIRE(((flag) & 3) == 0 || ((flag) & 3) == 3, "the stem takes fp32 tensors or activations AND gradients bfloat16 (bf16-compute path)"); \
    if (((flag) & 3) == 3) { using ActT [[maybe_unused]] = ::ttk::bf16_t; using GradT [[maybe_unused]] = ::ttk::bf16_t; __VA_ARGS__; }              \
    else { using ActT [[maybe_unused]] = float; using GradT [[maybe_unused]] = float; __VA_ARGS__; }                                                \
  } while (0)

// The "apply on load" forms of BatchNorm (see ttk.h).  Every layer owns one block
// bn[TTK_BN_ROWS][C] of per-channel constants.  Both forms SUBTRACT FIRST:
//   forward : a  = max(scale*(y - mean) + beta (+skip), 0)
//   backward: dy = ga*(g - gmean) + gb*(y - mean)
// (folding them into one fma per tensor, scale*y + shift / cA*g + cB*y + cC, cancels catastrophically
// in fp32 when a channel's values are nearly constant over the batch - which real loss gradients are).
struct BnApply4 {
  float4 scale, mean, beta;
  __device__ __forceinline__ static BnApply4 load(const float* bn, int C, int c) {
    return BnApply4{ld4(bn + TTK_BN_SCALE * C + c), ld4(bn + TTK_BN_MEAN * C + c), ld4(bn + TTK_BN_BETA * C + c)};
  }
  __device__ __forceinline__ float4 pre(float4 y) const { return fma4(scale, sub4(y, mean), beta); }
  __device__ __forceinline__ float4 act(float4 y) const { return relu4(pre(y)); }
  __device__ __forceinline__ float4 act(float4 y, float4 skip) const { return relu4(add4(pre(y), skip)); }
};
struct BnGrad4 {
  float4 ga, gb, gmean, mean;
  __device__ __forceinline__ static BnGrad4 load(const float* bn, int C, int c) {
    return BnGrad4{ld4(bn + TTK_BN_GA * C + c), ld4(bn + TTK_BN_GB * C + c), ld4(bn + TTK_BN_GMEAN * C + c),
                   ld4(bn + TTK_BN_MEAN * C + c)};
  }
  __device__ __forceinline__ float4 dy(float4 g, float4 y) const {
    return fma4(ga, sub4(g, gmean), mul4(gb, sub4(y, mean)));
  }
};

// ---- per-channel partial sums of a workgroup -------------------------------------------------
// Every thread of the block owns channel quad `c4` (4 consecutive channels starting at 4*c4) and
// has accumulated s1/s2 over its items.  Lanes that own the same quad are folded with wave
// shuffles (they sit `lanes_per_item` apart), then the waves meet in LDS; the block writes ONE row
// part[row][2][C].  Deterministic: no global atomics.
template <int MAXC, int NT = kBlock>
__device__ __forceinline__ void block_channel_partials(float4 s1, float4 s2, int c4, int C, float* part_row,
                                                        float* smem /* [2*MAXC] */) {
  const int tid = threadIdx.x;
  const int quads = C >> 2;  // threads per item (power of two, 8..256)
  // fold lanes of one wave that own the same quad: strides quads, 2*quads, ... < 64
  for (int off = quads; off < kWave; off <<= 1) {
    s1.x += __shfl_xor(s1.x, off); s1.y += __shfl_xor(s1.y, off);
    s1.z += __shfl_xor(s1.z, off); s1.w += __shfl_xor(s1.w, off);
    s2.x += __shfl_xor(s2.x, off); s2.y += __shfl_xor(s2.y, off);
    s2.z += __shfl_xor(s2.z, off); s2.w += __shfl_xor(s2.w, off);
  }
  for (int i = tid; i < 2 * C; i += NT) smem[i] = 0.f;
  __syncthreads();
  const int lane = tid & (kWave - 1);
  const bool owner = (quads >= kWave) || (lane < quads);
  // fixed-order accumulation (wave 0, then 1, ...): bitwise reproducible, no LDS atomics
  for (int w = 0; w < NT / kWave; ++w) {
    if ((tid >> 6) == w && owner) {
      float* d = smem + 4 * c4;
      d[0] += s1.x; d[1] += s1.y; d[2] += s1.z; d[3] += s1.w;
      d += C;
      d[0] += s2.x; d[1] += s2.y; d[2] += s2.z; d[3] += s2.w;
    }
    __syncthreads();
  }
  for (int i = tid; i < 2 * C; i += NT) part_row[i] = smem[i];
}

}  // namespace ttk
