// Depthwise 3x3 convolution (pad 1, stride 1|2) on channels-last fp32 activations, with the
// producer's BatchNorm+ReLU(+residual) applied while loading and this layer's BatchNorm statistics
// reduced in the epilogue.  Reference: DepthWiseBlock.conv_dw/bn_dw, backbones/mobilenet_v1.py:57-66,78-80.
//
// This file: the standalone depthwise weight gradient (unit tests; the training step uses the fused one
// in dwconv_tiled.hip) and the bn_act materialisation kernel.  Forward / data gradient: dwconv_tiled.hip.
// Thread = (pixel, channel quad); the C/4 lanes of one pixel read/write C*4 contiguous bytes.
// Channel counts are powers of two (32..1024) so quad ownership is fixed per thread, and the odd
// spatial sizes (65/33/17/9/5) never touch the lane mapping.
#include "ttk_common.h"

namespace ttk {

__global__ void zero_f(float* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

struct DwWeights {  // 3x3 filters of 4 consecutive channels: w[c][tap] as stored by the reference (C,1,3,3)
  float v[36];
  __device__ __forceinline__ void load(const float* w, int c4) {
    const float4* p = reinterpret_cast<const float4*>(w + 36 * c4);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      float4 q = p[i];
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
  }
  __device__ __forceinline__ float4 tap(int t) const { return make_float4(v[t], v[9 + t], v[18 + t], v[27 + t]); }
};

struct InputForm {  // how the block input is formed from the producer's raw output
  const float* yprev;
  const float* skip_prev;  // nullable
  const float* a_in;       // nullable: materialised input
  BnApply4 bn;
  __device__ __forceinline__ float4 operator()(size_t off) const {
    if (a_in) return ld4(a_in + off);
    const float4 y = ld4(yprev + off);
    return skip_prev ? bn.act(y, ld4(skip_prev + off)) : bn.act(y);
  }
};

// dW[c][tap] += sum_{n,ho,wo} dy_dw[n,ho,wo,c] * a_in[n, ho*S+kh-1, wo*S+kw-1, c]
template <int S>
__global__ void __launch_bounds__(kBlock) dw_bwd_weight_k(const float* __restrict__ g_dw, const float* __restrict__ y_dw,
                                                           const float* __restrict__ bn_dw, const float* __restrict__ yprev,
                                                           const float* __restrict__ bn_prev,
                                                           const float* __restrict__ skip_prev, const float* __restrict__ a_in,
                                                           float* __restrict__ dw, int B, int H, int W, int C, int Ho, int Wo,
                                                           int qshift) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [9][C]
  const int quads = C >> 2;
  const int c4 = threadIdx.x & (quads - 1);
  const BnGrad4 bg = BnGrad4::load(bn_dw, C, 4 * c4);
  InputForm in{yprev, skip_prev, a_in, BnApply4::load(bn_prev, C, 4 * c4)};
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f4(0.f);
  const int64_t items = ((int64_t)B * Ho * Wo) << qshift;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    int64_t pix = idx >> qshift;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const float4 dy = bg.dy(ld4(g_dw + (idx << 2)), ld4(y_dw + (idx << 2)));
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * S + kh - 1;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = wo * S + kw - 1;
        if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
        const size_t off = (((size_t)n * H + hi) * W + wi) * C + 4 * c4;
        acc[kh * 3 + kw] = fma4(dy, in(off), acc[kh * 3 + kw]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float4 v = acc[t];
    for (int off = quads; off < kWave; off <<= 1) {
      v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off);
      v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    }
    acc[t] = v;
  }
  for (int i = threadIdx.x; i < 9 * C; i += kBlock) smem[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const bool owner = (quads >= kWave) || (lane < quads);
  for (int wv = 0; wv < kBlock / kWave; ++wv) {
    if ((threadIdx.x >> 6) == wv && owner) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float* d = smem + t * C + 4 * c4;
        d[0] += acc[t].x; d[1] += acc[t].y; d[2] += acc[t].z; d[3] += acc[t].w;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 9 * C; i += kBlock) {
    const int t = i / C, c = i - t * C;
    atomicAdd(dw + c * 9 + t, smem[i]);
  }
}

// a = max(scale*y + shift (+skip), 0) over [rows][C]
__global__ void __launch_bounds__(kBlock) bn_act_k(const float* __restrict__ y, const float* __restrict__ bnp,
                                                    const float* __restrict__ skip,
                                                    float* __restrict__ a, int64_t items, int C) {
  const int quads = C >> 2;
  const int c4 = threadIdx.x & (quads - 1);
  const BnApply4 bn = BnApply4::load(bnp, C, 4 * c4);
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    const size_t off = (size_t)idx << 2;
    st4(a + off, skip ? bn.act(ld4(y + off), ld4(skip + off)) : bn.act(ld4(y + off)));
  }
}

static int log2i(int v) {
  int r = 0;
  while ((1 << r) < v) ++r;
  return r;
}
static bool dw_shape_ok(int B, int H, int W, int C, int stride) {
  return B > 0 && H > 0 && W > 0 && C >= 32 && C <= 1024 && (C & (C - 1)) == 0 && (stride == 1 || stride == 2);
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_dwconv3x3_bwd_weight(const float* g_dw, const float* y_dw, const float* bn_dw, const float* yprev, const float* bn_prev,
                             const float* skip_prev, const float* a_in, float* dw, int accumulate, int B, int H, int W, int C,
                             int stride, ttk_stream_t stream) {
  TTK_REQUIRE(g_dw && y_dw && bn_dw && yprev && bn_prev && dw, "dwconv3x3_bwd_weight: null pointer");
  TTK_REQUIRE(dw_shape_ok(B, H, W, C, stride), "dwconv3x3_bwd_weight: unsupported shape");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int qs = log2i(C / 4);
  const int64_t items = ((int64_t)B * Ho * Wo) << qs;
  int g = elementwise_grid(items);
  if (g > 256) g = 256;  // every block ends with 9*C float atomics: keep the count modest
  if (!accumulate) hipLaunchKernelGGL(zero_f, dim3((9 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, dw, (int64_t)9 * C);
  const size_t sm = 9 * (size_t)C * sizeof(float);
  if (stride == 1)
    hipLaunchKernelGGL(dw_bwd_weight_k<1>, dim3(g), dim3(kBlock), sm, (hipStream_t)stream, g_dw, y_dw, bn_dw, yprev, bn_prev,
                       skip_prev, a_in, dw, B, H, W, C, Ho, Wo, qs);
  else
    hipLaunchKernelGGL(dw_bwd_weight_k<2>, dim3(g), dim3(kBlock), sm, (hipStream_t)stream, g_dw, y_dw, bn_dw, yprev, bn_prev,
                       skip_prev, a_in, dw, B, H, W, C, Ho, Wo, qs);
  TTK_LAUNCH_CHECK("dwconv3x3_bwd_weight");
}

int ttk_bn_act(const float* y, const float* bn, const float* skip, float* a, int64_t rows, int C, ttk_stream_t stream) {
  TTK_REQUIRE(y && bn && a, "bn_act: null pointer");
  TTK_REQUIRE(rows > 0 && C >= 32 && C <= 1024 && (C & (C - 1)) == 0, "bn_act: unsupported shape rows=%lld C=%d", (long long)rows, C);
  const int64_t items = rows * (C / 4);
  int g = (int)ceil_div(items, kBlock);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(bn_act_k, dim3(g), dim3(kBlock), 0, (hipStream_t)stream, y, bn, skip, a, items, C);
  TTK_LAUNCH_CHECK("bn_act");
}

}  // extern "C"
