// Weight gradient of a stride-2 stem convolution with ONE input channel; used for the 7x7 -> 64 channel stem of the
// ResNet18 variant (backbones/resnet.py:63-66).  (Instantiated for the 5x5 -> 32 channel MobileNet stem it measured
// 278 us against 241 us of stem.hip's per-pixel kernel - rows of 65 x 32 values are too short for the staging - so
// stem.hip keeps its own.)
//   dW[c][tap] = sum_pixels dy[pixel][c] * x[tap of pixel],   dy = ga*(g-gmean) + gb*(y-mean) formed on load.
// A workgroup owns output rows (n, ho): the KS input rows and the dy row (transposed to [channel][pixel]) are staged in
// LDS once.  Thread (channel c, group grp) takes every G-th quad of 4 consecutive output pixels and accumulates ALL
// KS*KS taps for it in registers: one ds_read_b128 of dy and, per filter row, 3-4 wave-uniform ds_read_b128 of the
// input row feed 4*KS FMAs each.  History: v1 fetched the input values of every pixel as broadcast GLOBAL loads (a chain
// of latencies, 2.5 ms for the 7x7 stem at B=512); v2 staged through LDS but issued one ds_read_b32 per FMA and was
// LDS-instruction bound (0.9 ms).
#pragma once
#include "ttk_common.h"

namespace ttk {

template <int KS, int C>
__global__ void __launch_bounds__(kBlock) stem_wgrad_lds_k(const float* __restrict__ g, const float* __restrict__ y,
                                                            const float* __restrict__ bn, const float* __restrict__ x,
                                                            float* __restrict__ dw, float* __restrict__ partial, int B, int H, int W,
                                                            int Ho, int Wo, int Wp4, int Wo4) {
  constexpr int PAD = KS / 2, TAPS = KS * KS, G = kBlock / C;
  constexpr int NX = (6 + KS + 3) / 4;  // float4 loads covering x[8q .. 8q + 6 + KS - 1]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                 // [KS][Wp4]: xs[kh][wi + PAD], zero outside the image
  float* dyT = smem + KS * Wp4;     // [C][Wo4]: zero for wo >= Wo
  float* red = dyT + C * Wo4;       // [C][TAPS]
  const int c = threadIdx.x % C, grp = threadIdx.x / C;
  const float ga = bn[TTK_BN_GA * C + c], gb = bn[TTK_BN_GB * C + c], gmean = bn[TTK_BN_GMEAN * C + c], mean = bn[TTK_BN_MEAN * C + c];
  float acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) acc[t] = 0.f;
  const int nquads = (Wo + 3) / 4;
  const int64_t nrows = (int64_t)B * Ho;
  for (int64_t row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int ho = (int)(row % Ho), n = (int)(row / Ho);
    __syncthreads();  // previous row's readers are done
    for (int i = threadIdx.x; i < KS * Wp4; i += kBlock) {
      const int kh = i / Wp4, wi = i % Wp4 - PAD, hi = 2 * ho + kh - PAD;
      xs[i] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((size_t)n * H + hi) * W + wi] : 0.f;
    }
    const size_t rbase = (size_t)row * Wo * C;
    for (int i = threadIdx.x; i < Wo4 * C; i += kBlock) {  // i % C == c for every i (kBlock is a multiple of C)
      const int wo = i / C;
      dyT[c * Wo4 + wo] = wo < Wo ? fmaf(ga, g[rbase + i] - gmean, gb * (y[rbase + i] - mean)) : 0.f;
    }
    __syncthreads();
    for (int q = grp; q < nquads; q += G) {
      const float4 d4 = ld4(dyT + c * Wo4 + 4 * q);
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        float xv[4 * NX];
#pragma unroll
        for (int u = 0; u < NX; ++u) {
          const float4 v = ld4(xs + kh * Wp4 + 8 * q + 4 * u);
          xv[4 * u] = v.x; xv[4 * u + 1] = v.y; xv[4 * u + 2] = v.z; xv[4 * u + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) acc[kh * KS + kw] = fmaf(dv[i], xv[2 * i + kw], acc[kh * KS + kw]);
      }
    }
  }
  // fold the G pixel groups of each channel in a fixed order, then one atomicAdd per weight and workgroup
  __syncthreads();
  for (int gg = 0; gg < G; ++gg) {
    if (grp == gg) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) red[c * TAPS + t] = (gg == 0 ? 0.f : red[c * TAPS + t]) + acc[t];
    }
    __syncthreads();
  }
  // dw[c][tap] (zeroed by the caller) - or, deterministic mode, this workgroup's row of partial[grid][C * TAPS], folded in a
  // fixed order afterwards
  for (int i = threadIdx.x; i < C * TAPS; i += kBlock) {
    if (partial) partial[(size_t)blockIdx.x * C * TAPS + i] = red[i];
    else atomicAdd(dw + i, red[i]);
  }
}

inline int stem_wgrad_grid(int B, int Ho) {
  // few, persistent workgroups: every workgroup ends with one atomicAdd per weight, and same-address atomics serialise
  int64_t grid = (int64_t)B * Ho;
  return (int)(grid > 1024 ? 1024 : grid);
}

template <int KS, int C>
inline void launch_stem_wgrad(const float* g, const float* y, const float* bn, const float* x, float* dw, float* partial, int B, int H, int W,
                              int Ho, int Wo, hipStream_t st) {
  const int grid = stem_wgrad_grid(B, Ho);
  const int Wo4 = ((Wo + 3) / 4) * 4;
  const int Wp4 = ((2 * Wo4 + KS + 6 + 3) / 4) * 4 + 16;  // the last quad reads 4*NX floats from 8*q
  const size_t sm = (size_t)(KS * Wp4 + C * Wo4 + C * KS * KS) * sizeof(float);
  hipLaunchKernelGGL((stem_wgrad_lds_k<KS, C>), dim3((unsigned)grid), dim3(kBlock), sm, st, g, y, bn, x, dw, partial, B, H, W, Ho, Wo, Wp4, Wo4);
  if (partial) launch_fold_partials(partial, grid, (int64_t)C * KS * KS, dw, 1, st);
}

}  // namespace ttk
