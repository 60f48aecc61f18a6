// Shared helpers of the bf16-COMPUTE path (`--precision bf16-compute`, BASELINE config 5's bf16 leg; csrc/bc_*.hip).
//
// What differs from the fp32 path:
//  * every activation-sized tensor (raw conv outputs, materialised block inputs AND their gradients) is bf16 in CHANNEL BLOCKS OF 64:
//        [C / 64][M][64]   element (m, c) at ((c >> 6) * M + m) * 64 + (c & 63)     (C = 32: plain [M][32])
//    so a pixel of a block is ONE 128-byte line - the unit the fp32 path's 32-channel blocks have in fp32 - and a depthwise
//    workgroup's slab / a GEMM's k64 step are contiguous runs of pixels x 128 B;
//  * the pointwise GEMMs multiply bf16 operands in ONE v_mfma_f32_32x32x16_bf16 product with fp32 accumulation (no fp16 split, no scale
//    bounds, no conversion waves): the A operand is rounded to bf16 after BatchNorm + ReLU on load, the gradient operand after the
//    BatchNorm-backward form, the weights once per step (ttk_bc_prepare_weights);
//  * the depthwise kernels keep their LDS tiles in bf16 (a pixel of a 64-channel slab = 128 B there too) and accumulate in fp32.
// Master weights, BatchNorm statistics (taken from the values as stored), every reduction, the weight gradients and Adam stay fp32.
// With 8 mantissa bits in every stored tensor the one-fma forms of the BatchNorm maps (scale*y + shift; ga*g + gb*y + c0) are used:
// their fp32 cancellation error (2^-24 of the terms) is far below the storage quantisation (2^-9) that the subtract-first forms of the
// fp32 path protect.
#pragma once
#include "ttk_common.h"
#include <atomic>

namespace ttk {
namespace bc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// channels per block of a C-channel tensor, and the element offset of (m, c)
__host__ __device__ __forceinline__ int cbw(int C) { return C < 64 ? C : 64; }
__host__ __device__ __forceinline__ size_t off64(int64_t m, int c, int64_t M, int C) {
  const int w = cbw(C);
  return ((size_t)(c / w) * (size_t)M + (size_t)m) * w + (c % w);
}

// 8 bf16 (one 16-byte chunk) <-> 8 floats
__device__ __forceinline__ void unpack8(uint4 u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {  // round to nearest even (v_cvt_pk_bf16_f32)
  const ttk_bf16x2 v = __builtin_convertvector(ttk_f32x2{a, b}, ttk_bf16x2);
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}
__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 ld16nt(const void* p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
// relu on packed bf16 pairs: a negative float has the sign bit set = a negative int16 (v_pk_max_i16)
__device__ __forceinline__ unsigned relu_pk(unsigned v) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 a = __builtin_bit_cast(s16x2, v), z = {0, 0};
  const s16x2 r = __builtin_elementwise_max(a, z);
  return __builtin_bit_cast(unsigned, r);
}

// ---- channel pairs in packed fp32 (v_pk_fma_f32) -------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
// widen a 16-byte chunk into four channel pairs
__device__ __forceinline__ void unpack_f2(uint4 u, f2 (&v)[4]) {
  v[0] = f2{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)};
  v[1] = f2{__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
  v[2] = f2{__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u)};
  v[3] = f2{__uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
}
__device__ __forceinline__ uint4 pack_f2(const f2 (&v)[4]) {
  return make_uint4(pack2(v[0].x, v[0].y), pack2(v[1].x, v[1].y), pack2(v[2].x, v[2].y), pack2(v[3].x, v[3].y));
}
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void ld8(const float* p, f2 (&v)[4]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = f2{a.x, a.y}; v[1] = f2{a.z, a.w}; v[2] = f2{b.x, b.y}; v[3] = f2{b.z, b.w};
}
// operand chunks of the pixel-contraction kernels (weight gradient, fused backward): 8 channels of one pixel
//   dy = bf16(ga*g + (gb*y + c0));   a = relu(bf16(scale*x + shift))  (= bf16(relu(.)): rounding keeps sign and zero)
__device__ __forceinline__ uint4 dy_chunk(u32x4 g, u32x4 y, const f2 (&ga)[4], const f2 (&gb)[4], const f2 (&c0)[4]) {
  f2 gg[4], yy[4];
  unpack_f2(make_uint4(g.x, g.y, g.z, g.w), gg);
  unpack_f2(make_uint4(y.x, y.y, y.z, y.w), yy);
#pragma unroll
  for (int k = 0; k < 4; ++k) gg[k] = fma2(ga[k], gg[k], fma2(gb[k], yy[k], c0[k]));
  return pack_f2(gg);
}
__device__ __forceinline__ uint4 act_chunk(u32x4 x, const f2 (&sc)[4], const f2 (&sh)[4]) {
  f2 xx[4];
  unpack_f2(make_uint4(x.x, x.y, x.z, x.w), xx);
#pragma unroll
  for (int k = 0; k < 4; ++k) xx[k] = fma2(sc[k], xx[k], sh[k]);
  uint4 p = pack_f2(xx);
  p.x = relu_pk(p.x); p.y = relu_pk(p.y); p.z = relu_pk(p.z); p.w = relu_pk(p.w);
  return p;
}

// kernels that declare more than 64 KB of dynamic LDS need the attribute - per kernel instantiation AND per device: one bit per device ordinal,
// set with a relaxed atomic (forward runs on the main thread, backward on autograd's: both may come here first; setting the attribute twice is
// harmless).  The call's result is returned so that a launch function can report it instead of failing later with an opaque launch error.
template <auto Kern>
inline hipError_t allow_big_lds() {
  static std::atomic<unsigned long long> done{0ull};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;  // (ordinals beyond the mask: set it every time)
  const unsigned long long bit = 1ull << dev;
  if (dev != 63 && (done.load(std::memory_order_relaxed) & bit)) return hipSuccess;
  const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (rc == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
  return rc;
}

// out[i] (+)= partial[0][i] + partial[1][i] + ...  for FEW outputs and MANY rows (the depthwise weight gradient's workgroup rows, the slice
// partials of the small pointwise layers): 16 float4 columns x 64 row groups per 1024-thread block - a thread adds rows / 64 rows (loads in
// flight together), the groups are folded by a fixed tree in LDS (bitwise reproducible).  The plain one-thread-per-output folds took
// 9 - 13 us per launch on these shapes (latency of hundreds of dependent loads), 25 launches per step.
__device__ __forceinline__ void fold_rows_fast_body(const float* __restrict__ partial, int rows, int64_t n, float* __restrict__ out, int accumulate, unsigned block) {
  __shared__ float4 sm[64][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t i = ((int64_t)block * 16 + tx) * 4;
  float4 a = f4(0.f);
  if (i < n) {
    int r = ty;
    for (; r + 3 * 64 < rows; r += 4 * 64) {
      const float4 v0 = ld4nt(partial + (size_t)r * n + i), v1 = ld4nt(partial + (size_t)(r + 64) * n + i);
      const float4 v2 = ld4nt(partial + (size_t)(r + 128) * n + i), v3 = ld4nt(partial + (size_t)(r + 192) * n + i);
      a = add4(add4(add4(add4(a, v0), v1), v2), v3);
    }
    for (; r < rows; r += 64) a = add4(a, ld4nt(partial + (size_t)r * n + i));
  }
  sm[ty][tx] = a;
  __syncthreads();
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    if (ty < step) sm[ty][tx] = add4(sm[ty][tx], sm[ty + step][tx]);
    __syncthreads();
  }
  if (ty == 0 && i < n) {
    float4 t = sm[0][tx];
    if (accumulate) t = add4(t, ld4(out + i));
    st4(out + i, t);
  }
}
// out[i] += partial[0][i] + partial[1][i] + ... for MANY outputs (the slice tiles of the wide pointwise weight gradients: up to 1 M outputs, <= 64 rows):
// 16 B per thread, eight loads in flight, rows in order.  `block` of `nthreads` threads.
__device__ __forceinline__ void fold_rows_wide_body(const float* __restrict__ partial, int rows, int64_t n, float* __restrict__ out, unsigned block, int nthreads) {
  const int64_t i = ((int64_t)block * nthreads + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = ld4(out + i);
  int s = 0;
  for (; s + 8 <= rows; s += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld4nt(partial + (size_t)(s + u) * n + i);
#pragma unroll
    for (int u = 0; u < 8; ++u) a = add4(a, v[u]);
  }
  for (; s < rows; ++s) a = add4(a, ld4nt(partial + (size_t)s * n + i));
  st4(out + i, a);
}
template <int kDummy = 0>
__global__ void __launch_bounds__(1024) fold_rows_fast_k(const float* __restrict__ partial, int rows, int64_t n, float* __restrict__ out, int accumulate) {
  fold_rows_fast_body(partial, rows, n, out, accumulate, blockIdx.x);
}
inline bool fold_rows_fast_ok(int rows, int64_t n) { return n % 4 == 0 && n <= 65536 && rows >= 16; }
inline unsigned fold_rows_fast_blocks(int64_t n) { return (unsigned)((n / 4 + 15) / 16); }
inline bool launch_fold_rows_fast(const float* partial, int rows, int64_t n, float* out, int accumulate, hipStream_t st) {
  if (!fold_rows_fast_ok(rows, n)) return false;
  hipLaunchKernelGGL(fold_rows_fast_k<0>, dim3(fold_rows_fast_blocks(n)), dim3(1024), 0, st, partial, rows, n, out, accumulate);
  return true;
}

// Weight image of a pointwise layer for the GEMM kernels (ttk_bc_prepare_weights): for the product out[n] = sum_k in[k] * Wimg(n, k)
// rows n of min(K, 64) bf16 (one k64 block, 128 B; K = 32: 64 B) in [K / 64][N][row]; the 16-byte chunks of a row are stored XOR-swizzled
// so that the MFMA operand reads (32 rows x one chunk per ds_read_b128) are bank-conflict free in LDS:
//   chunk c (k = 8 c .. 8 c + 7 of the block) of row n sits at position c ^ swz(n)
__host__ __device__ __forceinline__ int w_cpr(int K) { return K < 64 ? K / 8 : 8; }                  // chunks per row
__host__ __device__ __forceinline__ int w_swz(int n, int cpr) { return cpr == 8 ? (n >> 1) & 7 : (n >> 2) & 3; }

constexpr int kFwd = 0, kDgrad = 1;

// constants of the output side, channels n0 .. n0 + cnt - 1: forward pivot; data gradient scale | shift | mean of the mask operand's BatchNorm
template <int MODE>
__device__ __forceinline__ void fill_cE(float* cE, const float* pivot, const float* bnE, int Ntot, int n0, int cnt, int tid, int nthreads) {
  for (int c = tid; c < cnt; c += nthreads) {
    if constexpr (MODE == kFwd) {
      cE[c] = pivot ? pivot[n0 + c] : 0.f;
    } else {
      const float sc = bnE[TTK_BN_SCALE * Ntot + n0 + c], mu = bnE[TTK_BN_MEAN * Ntot + n0 + c];
      cE[c] = sc;
      cE[cnt + c] = fmaf(-sc, mu, bnE[TTK_BN_BETA * Ntot + n0 + c]);
      cE[2 * cnt + c] = mu;
    }
  }
}

// One 32-pixel x OC-channel block of a wave's accumulators -> its LDS tile -> global memory (+ statistics).
//   acc: BPC = OC / 32 accumulator blocks (channels x pixels); stg: wave-private tile [32 px][OC] bf16, chunks XOR-swizzled by pixel
//   ce: output-side constants of THIS block's channels ([cnt] rows apart), pix0: first pixel, pend: end of the valid pixels
//   s1, s2: the lane's running sums for its 8 channels (lane % LPP = its 16-byte chunk of the pixel)
template <int MODE, int OC>
__device__ __forceinline__ void store_block(const f32x16* acc, uint4* stg, bf16_t* __restrict__ outb, const uint4* mk, const float* ce, int cnt,
                                            int64_t pix0, int64_t pend, float (&s1)[8], float (&s2)[8]) {
  constexpr int LPP = OC / 8, PPI = 64 / LPP, NI = 32 / PPI, BPC = OC / 32;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  unsigned char* sb = reinterpret_cast<unsigned char*>(stg);
#pragma unroll
  for (int blk = 0; blk < BPC; ++blk)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const f32x16& a = acc[blk];
      const uint2 v = make_uint2(pack2(a[4 * gq], a[4 * gq + 1]), pack2(a[4 * gq + 2], a[4 * gq + 3]));
      const int chunk = (4 * blk + gq) ^ (r & (LPP - 1));
      *reinterpret_cast<uint2*>(sb + r * (OC * 2) + chunk * 16 + 8 * h) = v;
    }
  asm volatile("" ::: "memory");  // (LDS operations of one wave complete in order: the reads below see the stores above)
  __builtin_amdgcn_wave_barrier();
  const int oct = lane % LPP;
  float e0[8], e1[8], e2[8];
  *reinterpret_cast<float4*>(e0) = *reinterpret_cast<const float4*>(ce + 8 * oct);
  *reinterpret_cast<float4*>(e0 + 4) = *reinterpret_cast<const float4*>(ce + 8 * oct + 4);
  if constexpr (MODE == kDgrad) {
    *reinterpret_cast<float4*>(e1) = *reinterpret_cast<const float4*>(ce + cnt + 8 * oct);
    *reinterpret_cast<float4*>(e1 + 4) = *reinterpret_cast<const float4*>(ce + cnt + 8 * oct + 4);
    *reinterpret_cast<float4*>(e2) = *reinterpret_cast<const float4*>(ce + 2 * cnt + 8 * oct);
    *reinterpret_cast<float4*>(e2 + 4) = *reinterpret_cast<const float4*>(ce + 2 * cnt + 8 * oct + 4);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int px = PPI * i + lane / LPP;
    uint4 u = stg[px * LPP + (oct ^ (px & (LPP - 1)))];
    const int64_t pix = pix0 + px;
    const bool ok = pix < pend;
    float f[8];
    unpack8(u, f);
    if constexpr (MODE == kFwd) {
      if (ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = f[j] - e0[j];
          s1[j] += d;
          s2[j] = fmaf(d, d, s2[j]);
        }
      }
    } else {
      float y[8];
      unpack8(mk[i], y);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = fmaf(e0[j], y[j], e1[j]);
        f[j] = a > 0.f ? f[j] : 0.f;
      }
      u = pack8(f);  // (the kept values are bf16 already: exact)
      if (ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s1[j] += f[j];
          s2[j] = fmaf(f[j], y[j] - e2[j], s2[j]);
        }
      }
    }
    if (ok) st16(outb + (size_t)(pix - pix0) * OC + 8 * oct, u);
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}


// ---- transposed fragments (the contraction runs over PIXELS: weight gradient) ----------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int wg_pitch(int T) { return T == 32 ? 64 : 2 * T + 64; }  // bytes of one pixel row of a T-channel plane (= 64 mod 256)

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* plane, int off, int pitch) {
  // pixels 8h .. 8h+3 and 8h+4 .. 8h+7 of the k16 step (the lane's address already holds 8h + q): element j = the lane's channel at pixel 8h + j
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + off + 4 * pitch));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

}  // namespace bc
}  // namespace ttk
