// Stem convolution 5x5, stride 2, pad 2, 1 -> 32 channels (backbones/mobilenet_v1.py:122-124,161)
// and its weight gradient.  HBM-bound: the output (B*65*65*32 floats) dominates; the 129x129 input
// plane is read through L1/L2 (each pixel is touched by <= 9 outputs x 8 lanes).
#include "ttk_common.h"

namespace ttk {

constexpr int kStemC = 32;
constexpr int kStemQuads = kStemC / 4;

// thread = TWO horizontally adjacent output pixels, all 32 channels.  The filter bank sits in LDS as
// wt[tap][c]; a ds_read_b128 at a wave-uniform address is a broadcast and hands every lane the weights of 4
// channels for one tap, which feed 8 FMAs (2 pixels x 4 channels).  35 coalesced input loads (5 rows x 7
// columns, shared by the two pixels) -> 1600 FMAs -> 2 x 128 B of output per thread.
// v1 (thread = pixel x channel quad: one LDS read per 4 FMAs, inputs re-loaded by 8 lanes) ran at 0.76 TB/s;
// a scalar-register filter bank does not work either (800 SGPRs: the compiler spills them through
// v_writelane/v_readlane).  BatchNorm partial sums stay in registers over the grid-stride loop.
__global__ void __launch_bounds__(kBlock) stem_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ y, float* __restrict__ part, int B, int H,
                                                      int W, int Ho, int Wo) {
  __shared__ __attribute__((aligned(16))) float wt[25][kStemC];
  __shared__ float red[kBlock / kWave][2 * kStemC];
  for (int i = threadIdx.x; i < 25 * kStemC; i += kBlock) wt[i % 25][i / 25] = w[i];  // w[c][tap] -> wt[tap][c]
  __syncthreads();
  float s1[kStemC], s2[kStemC];
#pragma unroll
  for (int c = 0; c < kStemC; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  const int Wpairs = (Wo + 1) / 2;
  const int64_t npairs = (int64_t)B * Ho * Wpairs;
  for (int64_t pr = (int64_t)blockIdx.x * kBlock + threadIdx.x; pr < npairs; pr += (int64_t)gridDim.x * kBlock) {
    // the filter bank is loop-invariant: without this the compiler hoists all 200 ds_read_b128 (800 VGPRs) out
    // of the pixel loop and spills.  The clobber pins the LDS reads inside the iteration.
    asm volatile("" ::: "memory");
    const int wp = (int)(pr % Wpairs), ho = (int)((pr / Wpairs) % Ho), n = (int)(pr / ((int64_t)Wpairs * Ho));
    const int wo = 2 * wp;
    const bool second = wo + 1 < Wo;
    const float* xn = x + (size_t)n * H * W;
    float xin[5][7];
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      const int hi = 2 * ho + kh - 2;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int wi = 2 * wo + k - 2;
        xin[kh][k] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? xn[(size_t)hi * W + wi] : 0.f;
      }
    }
    float* y0 = y + (((size_t)n * Ho + ho) * Wo + wo) * kStemC;
#pragma unroll
    for (int c4 = 0; c4 < kStemQuads; ++c4) {
      float4 a0 = f4(0.f), a1 = f4(0.f);
#pragma unroll
      for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
          const float4 wq = ld4(&wt[kh * 5 + kw][4 * c4]);
          a0 = fma4(f4(xin[kh][kw]), wq, a0);
          a1 = fma4(f4(xin[kh][kw + 2]), wq, a1);
        }
      st4(y0 + 4 * c4, a0);
      if (second) st4(y0 + kStemC + 4 * c4, a1);
      else a1 = f4(0.f);
      s1[4 * c4 + 0] += a0.x + a1.x; s1[4 * c4 + 1] += a0.y + a1.y; s1[4 * c4 + 2] += a0.z + a1.z; s1[4 * c4 + 3] += a0.w + a1.w;
      s2[4 * c4 + 0] = fmaf(a0.x, a0.x, fmaf(a1.x, a1.x, s2[4 * c4 + 0]));
      s2[4 * c4 + 1] = fmaf(a0.y, a0.y, fmaf(a1.y, a1.y, s2[4 * c4 + 1]));
      s2[4 * c4 + 2] = fmaf(a0.z, a0.z, fmaf(a1.z, a1.z, s2[4 * c4 + 2]));
      s2[4 * c4 + 3] = fmaf(a0.w, a0.w, fmaf(a1.w, a1.w, s2[4 * c4 + 3]));
    }
  }
  if (part) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < kStemC; ++c) {
      float a = s1[c], b = s2[c];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
      if (lane == 0) { red[wv][c] = a; red[wv][kStemC + c] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kStemC) {
      float a = 0.f;
      for (int i = 0; i < kBlock / kWave; ++i) a += red[i][threadIdx.x];  // fixed wave order
      part[(size_t)blockIdx.x * 2 * kStemC + threadIdx.x] = a;
    }
  }
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
    const float4 dy = bg.dy(ld4nt(g + (idx << 2)), ld4nt(y + (idx << 2)));
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
  const int64_t items = (int64_t)B * Ho * Wo * kStemQuads;  // sizes the partial rows (ttk_partial_rows_elementwise)
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
