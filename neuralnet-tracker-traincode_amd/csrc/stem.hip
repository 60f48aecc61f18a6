// Stem convolution 5x5, stride 2, pad 2, 1 -> 32 channels (backbones/mobilenet_v1.py:122-124,161)
// and its weight gradient.  HBM-bound: the output (B*65*65*32 floats) dominates; the 129x129 input
// plane is read through L1/L2 (each pixel is touched by <= 9 outputs x 8 lanes).
#include "ttk_common.h"

namespace ttk {

constexpr int kStemC = 32;
constexpr int kStemQuads = kStemC / 4;

// thread = (output pixel, channel quad); 8 consecutive lanes share one pixel and write its 128 B.
__global__ void __launch_bounds__(kBlock) stem_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ y, float* __restrict__ part, int B, int H,
                                                      int W, int Ho, int Wo) {
  __shared__ float wt[25][kStemC];  // transposed filter bank: wt[tap][c]
  __shared__ float red[2 * kStemC];
  for (int i = threadIdx.x; i < 25 * kStemC; i += kBlock) wt[i % 25][i / 25] = w[i];  // w[c][tap]
  __syncthreads();
  const int c4 = threadIdx.x & (kStemQuads - 1);
  const int64_t items = (int64_t)B * Ho * Wo * kStemQuads;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    int64_t pix = idx >> 3;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const float* xn = x + (size_t)n * H * W;
    float4 acc = f4(0.f);
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      const int hi = 2 * ho + kh - 2;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int wi = 2 * wo + kw - 2;
        if (wi < 0 || wi >= W) continue;
        const float v = xn[(size_t)hi * W + wi];
        acc = fma4(f4(v), ld4(&wt[kh * 5 + kw][4 * c4]), acc);
      }
    }
    st4(y + (idx << 2), acc);
    s1 = add4(s1, acc);
    s2 = fma4(acc, acc, s2);
  }
  if (part) block_channel_partials<kStemC>(s1, s2, c4, kStemC, part + (size_t)blockIdx.x * 2 * kStemC, red);
}

// dW[c][tap] = sum_{n,ho,wo} dy[n,ho,wo,c] * x[n, 2ho+kh-2, 2wo+kw-2]
// thread = (output pixel, channel quad) as in forward; 25 taps x 4 channels of partial sums per
// thread are folded over the workgroup through LDS, then one atomic add per (block, weight).
__global__ void __launch_bounds__(kBlock) stem_bwd_weight_k(const float* __restrict__ g, const float* __restrict__ y,
                                                             const float* __restrict__ bn, const float* __restrict__ x,
                                                             float* __restrict__ dw, int B, int H, int W, int Ho, int Wo) {
  __shared__ float acc_s[25][kStemC];
  for (int i = threadIdx.x; i < 25 * kStemC; i += kBlock) (&acc_s[0][0])[i] = 0.f;
  __syncthreads();
  const int c4 = threadIdx.x & (kStemQuads - 1);
  const BnGrad4 bg = BnGrad4::load(bn, kStemC, 4 * c4);
  float4 acc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = f4(0.f);
  const int64_t items = (int64_t)B * Ho * Wo * kStemQuads;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    int64_t pix = idx >> 3;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const float* xn = x + (size_t)n * H * W;
    const float4 dy = bg.dy(ld4(g + (idx << 2)), ld4(y + (idx << 2)));
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      const int hi = 2 * ho + kh - 2;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int wi = 2 * wo + kw - 2;
        const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
        const float v = ok ? xn[(size_t)hi * W + wi] : 0.f;
        acc[kh * 5 + kw] = fma4(f4(v), dy, acc[kh * 5 + kw]);
      }
    }
  }
  // fold the 8 pixel-lanes groups of each wave (lanes sharing c4 are 8 apart), then LDS
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    float4 v = acc[t];
    for (int off = kStemQuads; off < kWave; off <<= 1) {
      v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off);
      v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    }
    acc[t] = v;
  }
  const int lane = threadIdx.x & (kWave - 1);
  for (int wv = 0; wv < kBlock / kWave; ++wv) {
    if ((threadIdx.x >> 6) == wv && lane < kStemQuads) {
#pragma unroll
      for (int t = 0; t < 25; ++t) {
        float* d = &acc_s[t][4 * c4];
        d[0] += acc[t].x; d[1] += acc[t].y; d[2] += acc[t].z; d[3] += acc[t].w;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 25 * kStemC; i += kBlock) atomicAdd(dw + i, acc_s[i % 25][i / 25]);  // dw[c][tap]
}

__global__ void zero_k(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_stem_fwd(const float* x, const float* w, float* y, float* part, int B, int H, int W, ttk_stream_t stream) {
  TTK_REQUIRE(x && w && y, "stem_fwd: null pointer");
  TTK_REQUIRE(B > 0 && H > 4 && W > 4, "stem_fwd: bad shape B=%d H=%d W=%d", B, H, W);
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t items = (int64_t)B * Ho * Wo * kStemQuads;
  hipLaunchKernelGGL(stem_fwd_k, dim3(elementwise_grid(items)), dim3(kBlock), 0, (hipStream_t)stream, x, w, y, part, B, H,
                     W, Ho, Wo);
  TTK_LAUNCH_CHECK("stem_fwd");
}

int ttk_stem_bwd_weight(const float* g, const float* y, const float* bn, const float* x, float* dw, int accumulate, int B,
                        int H, int W, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn && x && dw, "stem_bwd_weight: null pointer");
  TTK_REQUIRE(B > 0 && H > 4 && W > 4, "stem_bwd_weight: bad shape");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t items = (int64_t)B * Ho * Wo * kStemQuads;
  if (!accumulate) hipLaunchKernelGGL(zero_k, dim3(4), dim3(256), 0, (hipStream_t)stream, dw, 25 * kStemC);
  int grid = elementwise_grid(items);
  if (grid > 512) grid = 512;
  hipLaunchKernelGGL(stem_bwd_weight_k, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, g, y, bn, x, dw, B, H, W,
                     Ho, Wo);
  TTK_LAUNCH_CHECK("stem_bwd_weight");
}

}  // extern "C"
