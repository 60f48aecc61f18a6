// Pointwise 1x1 convolutions of the NARROW, HBM-bound layers (N = 64 or 128 output channels, contraction K <= 256: the layers with the largest
// activations) - forward and data gradient - as a STREAMING kernel on the fp16 matrix pipe (round 6).  Reference: DepthWiseBlock.conv_sep + bn_sep,
// backbones/mobilenet_v1.py:67-68,82-84.  Arithmetic, operand bounds and numerics are those of pwconv_f16.hip (two fp16 pieces per operand, three
// v_mfma_f32_16x16x32_f16 per product, fp32 accumulation).
//
// These layers move 0.4 - 0.8 GB per launch for 5 - 18 GFLOP: the bytes, not the products, set their time, and the tiled kernels that ran them
// (pw16_k 256 x 128 tiles: 2 178 one-tile workgroups, each a prologue + 2-4 k32 steps behind barriers + an epilogue; pw_gemm_k for 32 -> 64)
// reached 4.2 - 4.95 TB/s where the depthwise kernels of the same tensors reach 5.3 - 6.0.  Here (the structure of the bf16 path's bc_gemm_e_k):
//  * the WHOLE weight operand (fragment-ordered fp16 planes, ttk_pwconv_prepare_weights split code 3: 16 - 128 KB) is resident in LDS;
//  * a workgroup is eight INDEPENDENT waves, persistent over groups of 16 PB pixels; after the prologue there is no barrier;
//  * the weights are the MFMA "A" operand (rows = output channels), the pixels the columns: lane (r, q) of a 16 x 16 x 32 product holds channels
//    8 q .. 8 q + 7 of pixel r - 32 contiguous bytes of the [K/32][M][32] tensor.  A wave loads its fragments STRAIGHT from global memory (two
//    16-byte loads per lane, pixel block and k32 step: whole 128-byte lines), forms the BatchNorm map and cuts the two fp16 pieces in registers:
//    the activations never touch LDS;
//  * an accumulator lane holds 4 consecutive channels of one pixel: 16-byte stores straight from the accumulators;
//  * BatchNorm partial sums: a group's sums are folded over the 16 pixel lanes by shuffles and kept by ONE owner lane per channel block
//    (8 registers per lane instead of 64), one row per workgroup at the end.
#include "ttk_common.h"
#include "conv_geom.h"
#include <atomic>
#include <type_traits>

namespace ttk {

typedef float yf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 yf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 yf16x2 __attribute__((ext_vector_type(2)));
typedef float yf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned yu32x4 __attribute__((ext_vector_type(4)));

enum { YMODE_FWD = 0, YMODE_DGRAD = 1 };

template <int N, typename F, int I = 0>
__device__ __forceinline__ void yfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    yfor<N, F, I + 1>(static_cast<F&&>(f));
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ybuf(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ yf32x4 yld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {  // streaming (nt)
  return __builtin_bit_cast(yf32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 2));
}
// fp16(x - h) of a pair (v_fma_mixlo/hi_f16), followed by the wait states a VALU result needs before an MFMA may read it: the result feeds an MFMA
// DIRECTLY here and hipcc pads nothing between an asm statement's VALU write and its consumer - without the s_nop the low pieces of one block came back
// wrong (1.5e-5 instead of 2e-7).  (The plain-C form - convert back, subtract, convert - lets hipcc hoist every conversion of a group to the front:
// 200 - 870 spilled registers, 2 - 4 x the time; profiles/r06_streaming_gemm.txt.)
__device__ __forceinline__ unsigned ylow2(unsigned h, float x0, float x1) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 3"
      : "=&v"(l)
      : "v"(h), "v"(x0), "v"(x1));
  return l;
}
__device__ __forceinline__ unsigned yhigh2(float x0, float x1) {
  const yf16x2 h = __builtin_convertvector(yf32x2{x0, x1}, yf16x2);
  return __builtin_bit_cast(unsigned, h);
}

// NCB: 16-channel blocks of the output (N = 16 NCB); NKS: k32 steps (K = 32 NKS); PB: 16-pixel blocks per group and wave
template <int MODE, int NCB, int NKS, int PB>
__global__ void __launch_bounds__(512, 2) pw16y_k(const float* __restrict__ A0, const float* __restrict__ A1, const float* __restrict__ bnA,
                                                  const uint16_t* __restrict__ Wq, const float* __restrict__ wmax, float* __restrict__ out,
                                                  const float* __restrict__ E0, const float* __restrict__ bnE, float* __restrict__ part, int64_t M) {
  constexpr bool FWD = MODE == YMODE_FWD;
  constexpr int K = 32 * NKS, N = 16 * NCB, PG = 16 * PB, RL = FWD ? 2 : 4, NCONST = FWD ? 3 : 4;
  constexpr int WBYTES = NKS * NCB * 2 * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* const cst = reinterpret_cast<float*>(lds + WBYTES);              // [NCONST][K]
  float* const red = cst + NCONST * K;                                    // [8 waves][2][N] at the end
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const float sa = pow2_scale(bnA[(size_t)TTK_BN_AUX * K + (FWD ? TTK_AUX_ACT_BOUND : TTK_AUX_DY_BOUND)]);
  const float sb = pow2_scale(*wmax);
  const float inv = 1.f / (sa * sb);  // exact: a power of two

  // ---- prologue: the weight image and the per-channel constants of the A operand (S_a folded in) into LDS
  {
    const uint4* src = reinterpret_cast<const uint4*>(Wq);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int i = tid; i < WBYTES / 16; i += 512) dst[i] = src[i];
    for (int i = tid * 4; i < NCONST * K; i += 512 * 4) {
      const int j = i / K, c = i - j * K;
      const int row = FWD ? (j == 0 ? TTK_BN_SCALE : (j == 1 ? TTK_BN_MEAN : TTK_BN_BETA)) : (j == 0 ? TTK_BN_GA : (j == 1 ? TTK_BN_GMEAN : (j == 2 ? TTK_BN_GB : TTK_BN_MEAN)));
      float4 v = ld4(bnA + (size_t)row * K + c);
      if (j == 0 || j == 2) v = make_float4(v.x * sa, v.y * sa, v.z * sa, v.w * sa);
      st4(cst + i, v);
    }
  }
  __syncthreads();

  const int64_t ngroups = (M + PG - 1) / PG;
  const int64_t gstride = (int64_t)gridDim.x * 8;
  // sums of this lane's OWN channel block (block fr & 7 when fr < NCB ... see the fold below): 4 channels x 2 sums
  yf32x4 os1 = yf32x4{0.f, 0.f, 0.f, 0.f}, os2 = os1;
  // epilogue constants of the lane's channels in every block would be 3 x 4 x NCB registers: read per block from global (L2) instead
  const unsigned char* const wrd = lds + lane * 16;

  yf32x4 raw[2][PB][RL];
  auto load_stage = [&](int64_t grp, auto ksc, auto setc) {  // requests the fragments of k32 step ks of pixel group grp (pixels clamped to M - 1)
    constexpr int ks = decltype(ksc)::value, set = decltype(setc)::value;
    const __amdgpu_buffer_rsrc_t r0 = ybuf(A0 + (size_t)ks * (size_t)M * 32);
    const __amdgpu_buffer_rsrc_t r1 = ybuf((FWD ? A0 : A1) + (size_t)ks * (size_t)M * 32);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      int64_t p = grp * PG + 16 * pb + fr;
      p = p < M ? p : M - 1;
      const unsigned off = (unsigned)p * 128u + (unsigned)fq * 32u;
      raw[set][pb][0] = yld16(r0, off);
      raw[set][pb][1] = yld16(r0, off + 16u);
      if constexpr (!FWD) {
        raw[set][pb][2] = yld16(r1, off);
        raw[set][pb][3] = yld16(r1, off + 16u);
      }
    }
  };

  int64_t grp = (int64_t)blockIdx.x * 8 + wave;
  if (grp < ngroups) load_stage(grp, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  for (; grp < ngroups; grp += gstride) {
    yf32x4 acc[NCB][PB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int p = 0; p < PB; ++p) acc[c][p] = yf32x4{0.f, 0.f, 0.f, 0.f};
    yfor<NKS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value, set = ks & 1;
      // the next stage's fragments are requested before this one is converted: the next k32 step of this group, or (last step) step 0 of the wave's
      // next group - into register set 0, which is free by then when NKS is even; with one k32 step it is the set being converted: see below
      const int64_t gn = grp + gstride < ngroups ? grp + gstride : grp;  // (the last group requests itself again: unconditional loads)
      if constexpr (ks + 1 < NKS) load_stage(grp, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, set ^ 1>{});
      else if constexpr ((NKS & 1) == 0) load_stage(gn, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      // constants of this lane's 8 channels of the step
      const float* cs = cst + ks * 32 + 8 * fq;
      yf32x4 cc[2 * NCONST];
#pragma unroll
      for (int i = 0; i < NCONST; ++i) {
        cc[2 * i] = *reinterpret_cast<const yf32x4*>(cs + i * K);
        cc[2 * i + 1] = *reinterpret_cast<const yf32x4*>(cs + i * K + 4);
      }
      yf16x8 bh[PB], bl[PB];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        yf32x4 t0, t1;
        if constexpr (FWD) {
          t0 = cc[0] * (raw[set][pb][0] - cc[2]) + cc[4];
          t1 = cc[1] * (raw[set][pb][1] - cc[3]) + cc[5];
#pragma unroll
          for (int j = 0; j < 4; ++j) { t0[j] = fmaxf(t0[j], 0.f); t1[j] = fmaxf(t1[j], 0.f); }
        } else {
          t0 = cc[0] * (raw[set][pb][0] - cc[2]) + cc[4] * (raw[set][pb][2] - cc[6]);
          t1 = cc[1] * (raw[set][pb][1] - cc[3]) + cc[5] * (raw[set][pb][3] - cc[7]);
        }
        yu32x4 h, l;
        h[0] = yhigh2(t0[0], t0[1]); h[1] = yhigh2(t0[2], t0[3]); h[2] = yhigh2(t1[0], t1[1]); h[3] = yhigh2(t1[2], t1[3]);
        l[0] = ylow2(h[0], t0[0], t0[1]); l[1] = ylow2(h[1], t0[2], t0[3]); l[2] = ylow2(h[2], t1[0], t1[1]); l[3] = ylow2(h[3], t1[2], t1[3]);
        bh[pb] = __builtin_bit_cast(yf16x8, h);
        bl[pb] = __builtin_bit_cast(yf16x8, l);
      }
      if constexpr (ks + 1 == NKS && (NKS & 1) == 1) load_stage(gn, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});  // (set 0 has just been consumed)
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        const yf16x8 wh = *reinterpret_cast<const yf16x8*>(wrd + ((ks * NCB + c) * 2 + 0) * 1024);
        const yf16x8 wl = *reinterpret_cast<const yf16x8*>(wrd + ((ks * NCB + c) * 2 + 1) * 1024);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          acc[c][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[pb], acc[c][pb], 0, 0, 0);
          acc[c][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[pb], acc[c][pb], 0, 0, 0);
          acc[c][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[pb], acc[c][pb], 0, 0, 0);
        }
      }
    });
    // ---- epilogue of the group: block (c, pb), lane (fr, fq): channels 16 c + 4 fq .. + 3 of pixel grp PG + 16 pb + fr
    const int64_t p0 = grp * PG;
    yfor<NCB>([&](auto cc_) {
      constexpr int c = decltype(cc_)::value;
      const int ch = 16 * c + 4 * fq;
      const size_t cbase = ((size_t)(ch >> 5) * (size_t)M) * 32 + (ch & 31);
      yf32x4 s1 = yf32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
      yf32x4 v[PB];
      if constexpr (FWD) {
        yf32x4 piv = yf32x4{0.f, 0.f, 0.f, 0.f};
        if (bnE) piv = *reinterpret_cast<const yf32x4*>(bnE + ch);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) v[pb] = acc[c][pb] * inv;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          const int64_t p = p0 + 16 * pb + fr;
          if (p < M) {
            *reinterpret_cast<yf32x4*>(out + cbase + (size_t)p * 32) = v[pb];
            const yf32x4 d = v[pb] - piv;
            s1 += d;
            s2 += d * d;
          }
        }
      } else {
        const yf32x4 esc = *reinterpret_cast<const yf32x4*>(bnE + (size_t)TTK_BN_SCALE * N + ch);
        const yf32x4 emu = *reinterpret_cast<const yf32x4*>(bnE + (size_t)TTK_BN_MEAN * N + ch);
        const yf32x4 ebe = *reinterpret_cast<const yf32x4*>(bnE + (size_t)TTK_BN_BETA * N + ch);
        yf32x4 ev[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          int64_t p = p0 + 16 * pb + fr;
          p = p < M ? p : M - 1;
          ev[pb] = __builtin_nontemporal_load(reinterpret_cast<const yf32x4*>(E0 + cbase + (size_t)p * 32));
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          const int64_t p = p0 + 16 * pb + fr;
          const yf32x4 yc = ev[pb] - emu;
          const yf32x4 a = esc * yc + ebe;
          yf32x4 t = acc[c][pb] * inv;
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = a[j] > 0.f ? t[j] : 0.f;
          v[pb] = t;
          if (p < M) {
            s1 += t;
            s2 += t * yc;
          }
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          const int64_t p = p0 + 16 * pb + fr;
          if (p < M) *reinterpret_cast<yf32x4*>(out + cbase + (size_t)p * 32) = v[pb];
        }
      }
      // fold over the 16 pixel lanes (every lane ends with the total), the lane with fr == c keeps it
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s1[j] += __shfl_xor(s1[j], off);
          s2[j] += __shfl_xor(s2[j], off);
        }
      }
      if (fr == c) { os1 += s1; os2 += s2; }
    });
  }
  // ---- one row of partial sums per workgroup: the eight waves in a fixed order (lane (fr = c, fq) owns channels 16 c + 4 fq .. + 3)
  __syncthreads();
  if (fr < NCB) {
    float* d = red + (size_t)wave * 2 * N + 16 * fr + 4 * fq;
    *reinterpret_cast<yf32x4*>(d) = os1;
    *reinterpret_cast<yf32x4*>(d + N) = os2;
  }
  __syncthreads();
  if (part && tid < 2 * N) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) a += red[(size_t)w * 2 * N + tid];
    part[(size_t)blockIdx.x * 2 * N + tid] = a;  // tid = which * N + channel
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
// Round 6's measurement (profiles/r06_streaming_gemm.txt): in the step this form is NOT faster than the tiled kernels it was to replace (32 -> 64 forward
// 188 us against 176, 64 -> 128 102.5 against 99.7, 128 -> 128 135.8 against 131.8, data gradient of 128 -> 256 143 against 106): these launches write
// two bytes for every byte they read and sit at the memory system's ceiling for that mix (the 1:1 copy probe of the same box: 5.4 TB/s), whatever the
// kernel's structure.  It stays an experiment build (TTK_GEMM_Y=1 with -DTTK_EXPERIMENTS); the product runs pw16_k / pw_gemm_k.
static bool f16y_enabled() {
  static const bool on = [] { const char* e = exp_env("TTK_GEMM_Y"); return e && e[0] == '1'; }();
  return on && gemm_mode() == GEMM_F16X2;
}
// forward: 32 -> 64, 64 -> 128, 128 -> 128; data gradient: of 128 -> 256 (K = 256, N = 128).  (The data gradients of the first three layers run in the
// fused backward kernels, pw_bwd_fused.hip.)
bool f16y_gemm_shape(int K, int Nout, int dgrad) {
  if (!f16y_enabled()) return false;
  if (!dgrad) return (K == 32 && Nout == 64) || ((K == 64 || K == 128) && Nout == 128);
  return K == 256 && Nout == 128;
}
static int y_grid(int64_t M, int pg) {
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int64_t groups = ceil_div(M, pg), wg = ceil_div(groups, 8);
  return (int)(wg < cus ? (wg < 1 ? 1 : wg) : cus);
}
template <int MODE, int NCB, int NKS, int PB>
static void y_launch(const float* A0, const float* A1, const float* bnA, const uint16_t* Wq, const float* wmax, float* out, const float* E0, const float* bnE,
                     float* part, int64_t M, hipStream_t st) {
  constexpr int K = 32 * NKS, N = 16 * NCB;
  const size_t sm = (size_t)NKS * NCB * 2048 + (size_t)(MODE == YMODE_FWD ? 3 : 4) * K * 4 + (size_t)8 * 2 * N * 4;
  static std::atomic<unsigned long long> done{0ull};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;
  if (dev == 63 || !(done.load(std::memory_order_relaxed) & (1ull << dev))) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pw16y_k<MODE, NCB, NKS, PB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess)
      done.fetch_or(1ull << dev, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((pw16y_k<MODE, NCB, NKS, PB>), dim3(y_grid(M, 16 * PB)), dim3(512), sm, st, A0, A1, bnA, Wq, wmax, out, E0, bnE, part, M);
}
constexpr int kYPbFwd = 3, kYPbDgrad = 2;
int f16y_partial_rows(int64_t M, int K, int Nout, int dgrad) { return f16y_gemm_shape(K, Nout, dgrad) ? y_grid(M, 16 * (dgrad ? kYPbDgrad : kYPbFwd)) : 0; }

__global__ void __launch_bounds__(256) w16y_split_k(const float* __restrict__ w, uint16_t* __restrict__ q, const float* __restrict__ wmax, int rows, int K) {
  const int64_t n = (int64_t)rows * K;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = pow2_scale(*wmax);
  const int row = (int)(i / K), k = (int)(i - (int64_t)row * K);
  const float x = w[i] * s;
  const _Float16 hh = (_Float16)x;
  const _Float16 ll = (_Float16)(x - (float)hh);
  q[x_plane_index(row, k, rows, 0)] = __builtin_bit_cast(uint16_t, hh);
  q[x_plane_index(row, k, rows, 1)] = __builtin_bit_cast(uint16_t, ll);
}
__global__ void __launch_bounds__(256) w16y_absmax_k(const float* __restrict__ w, int64_t n, unsigned* __restrict__ wmax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __hip_atomic_load(wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(wmax, __float_as_uint(m));
}

// Returns true when the shape was handled here.  `planes`: the fragment-ordered image (prepared block, split code 3), or - Bm != nullptr, the per-call
// form of the unit tests - scratch that is filled from the raw weight rows [Nout][K] first; wmax: the layer's |w| maximum.
template <int MODE>
bool launch_f16y_gemm(const float* A0, const float* A1, const float* bnA, const float* Bm, float* out, const float* E0, const float* bnE, float* part, int64_t M,
                      int K, int Nout, void* planes, float* wmax, hipStream_t st) {
  if (!planes || !wmax || !f16y_gemm_shape(K, Nout, MODE == YMODE_DGRAD) || M * 128 >= ((int64_t)1 << 32)) return false;
  if (Bm) {
    const int64_t nw = (int64_t)Nout * K;
    (void)hipMemsetAsync(wmax, 0, sizeof(float), st);
    hipLaunchKernelGGL(w16y_absmax_k, dim3((unsigned)(nw / 1024 < 1 ? 1 : (nw / 1024 > 256 ? 256 : nw / 1024))), dim3(256), 0, st, Bm, nw, reinterpret_cast<unsigned*>(wmax));
    hipLaunchKernelGGL(w16y_split_k, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, st, Bm, reinterpret_cast<uint16_t*>(planes), wmax, Nout, K);
  }
  const uint16_t* Wq = reinterpret_cast<const uint16_t*>(planes);
  if constexpr (MODE == YMODE_FWD) {
    if (K == 32 && Nout == 64) y_launch<MODE, 4, 1, kYPbFwd>(A0, A1, bnA, Wq, wmax, out, E0, bnE, part, M, st);
    else if (K == 64) y_launch<MODE, 8, 2, kYPbFwd>(A0, A1, bnA, Wq, wmax, out, E0, bnE, part, M, st);
    else y_launch<MODE, 8, 4, kYPbFwd>(A0, A1, bnA, Wq, wmax, out, E0, bnE, part, M, st);
  } else {
    y_launch<MODE, 8, 8, kYPbDgrad>(A0, A1, bnA, Wq, wmax, out, E0, bnE, part, M, st);
  }
  return true;
}
template bool launch_f16y_gemm<YMODE_FWD>(const float*, const float*, const float*, const float*, float*, const float*, const float*, float*, int64_t, int, int, void*,
                                          float*, hipStream_t);
template bool launch_f16y_gemm<YMODE_DGRAD>(const float*, const float*, const float*, const float*, float*, const float*, const float*, float*, int64_t, int, int, void*,
                                            float*, hipStream_t);

}  // namespace ttk
