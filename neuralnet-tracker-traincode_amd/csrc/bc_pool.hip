// Global average pooling over the last block's output and its backward on the bf16-COMPUTE path (bc_common.h): bf16 tensors in channel
// blocks of 64, 16-byte accesses (8 channels per lane), the producer's BatchNorm + residual + ReLU applied on load (one-fma form).
// Reference: AdaptiveAvgPool2d(1) + view, backbones/mobilenet_v1.py:143,180-181.
#include "bc_common.h"

namespace ttk {
namespace bc {

// thread = (sample, 8-channel chunk)
__global__ void __launch_bounds__(256) bc_avgpool_fwd_k(const bf16_t* __restrict__ y, const float* __restrict__ bnp, const bf16_t* __restrict__ skip,
                                                         float* __restrict__ feat, int B, int HW, int C) {
  const int chunks = C >> 3;
  const int64_t items = (int64_t)B * chunks, M = (int64_t)B * HW;
  const float inv = 1.0f / (float)HW;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % chunks) * 8, n = (int)(idx / chunks);
    float sc[8], sh[8], s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = bnp[TTK_BN_SCALE * C + c8 + j];
      sh[j] = fmaf(-sc[j], bnp[TTK_BN_MEAN * C + c8 + j], bnp[TTK_BN_BETA * C + c8 + j]);
      s[j] = 0.f;
    }
    const size_t base = off64((int64_t)n * HW, c8, M, C);
    for (int p = 0; p < HW; ++p) {
      float v[8], k[8];
      unpack8(ld16nt(y + base + (size_t)p * 64), v);
      if (skip) unpack8(ld16nt(skip + base + (size_t)p * 64), k);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = fmaf(sc[j], v[j], sh[j]);
        if (skip) a += k[j];
        s[j] += fmaxf(a, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) feat[(size_t)n * C + c8 + j] = s[j] * inv;
  }
}

// thread = (pixel slot, 8-channel chunk): the block strides over the pixels, a thread keeps its chunk; partial sums of the block in one row
__global__ void __launch_bounds__(256) bc_avgpool_bwd_k(const float* __restrict__ gfeat, const bf16_t* __restrict__ y, const float* __restrict__ bnp,
                                                         const bf16_t* __restrict__ skip, bf16_t* __restrict__ g, float* __restrict__ part, int B, int HW, int C) {
  extern __shared__ float red[];  // [slots][2][C]
  const int chunks = C >> 3, slots = 256 / chunks;
  const int ch = threadIdx.x % chunks, slot = threadIdx.x / chunks, c8 = ch * 8;
  const int64_t M = (int64_t)B * HW;
  const float inv = 1.0f / (float)HW;
  float sc[8], sh[8], mu[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = bnp[TTK_BN_SCALE * C + c8 + j];
    mu[j] = bnp[TTK_BN_MEAN * C + c8 + j];
    sh[j] = fmaf(-sc[j], mu[j], bnp[TTK_BN_BETA * C + c8 + j]);
    s1[j] = s2[j] = 0.f;
  }
  if (slot < slots) {
    for (int64_t m = (int64_t)blockIdx.x * slots + slot; m < M; m += (int64_t)gridDim.x * slots) {
      const int n = (int)(m / HW);
      const size_t off = off64(m, c8, M, C);
      float v[8], k[8], o[8];
      unpack8(ld16nt(y + off), v);
      if (skip) unpack8(ld16nt(skip + off), k);
      const float4 g0 = ld4(gfeat + (size_t)n * C + c8), g1 = ld4(gfeat + (size_t)n * C + c8 + 4);
      const float gf[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = fmaf(sc[j], v[j], sh[j]);
        if (skip) a += k[j];
        o[j] = a > 0.f ? gf[j] * inv : 0.f;
      }
      const uint4 pk = pack8(o);
      st16(g + off, pk);
      unpack8(pk, o);  // sums of what is stored
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s1[j] += o[j];
        s2[j] = fmaf(o[j], v[j] - mu[j], s2[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(slot * 2 + 0) * C + c8 + j] = s1[j];
      red[(slot * 2 + 1) * C + c8 + j] = s2[j];
    }
  }
  __syncthreads();
  if (part)
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
      const int which = i / C, c = i - which * C;
      float a = 0.f;
      for (int sl = 0; sl < slots; ++sl) a += red[(sl * 2 + which) * C + c];
      part[(size_t)blockIdx.x * 2 * C + i] = a;
    }
}

static int pool_bwd_grid(int64_t M, int C) {
  const int slots = 256 / (C / 8);
  int64_t g = ceil_div(M, (int64_t)slots * 4);
  if (g > TTK_MAX_PARTIAL_ROWS_ELEMENTWISE) g = TTK_MAX_PARTIAL_ROWS_ELEMENTWISE;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace bc
}  // namespace ttk

using namespace ttk;
using namespace ttk::bc;

extern "C" {

int ttk_bc_partial_rows_pool(int B, int HW, int C) { return (B > 0 && HW > 0 && C >= 64 && C <= 2048 && C % 64 == 0) ? pool_bwd_grid((int64_t)B * HW, C) : -1; }

int ttk_bc_avgpool_fwd(const void* y, const float* bn, const void* skip, float* feat, int B, int HW, int C, ttk_stream_t stream) {
  TTK_REQUIRE(y && bn && feat, "bc_avgpool_fwd: null pointer");
  TTK_REQUIRE(B > 0 && HW > 0 && C >= 64 && C <= 2048 && C % 64 == 0, "bc_avgpool_fwd: unsupported shape B=%d HW=%d C=%d", B, HW, C);
  int64_t grid = ceil_div((int64_t)B * (C / 8), 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(bc_avgpool_fwd_k, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, bn, (const bf16_t*)skip, feat, B, HW, C);
  TTK_LAUNCH_CHECK("bc_avgpool_fwd");
}

int ttk_bc_avgpool_bwd(const float* gfeat, const void* y, const float* bn, const void* skip, void* g, float* part, int B, int HW, int C,
                       ttk_stream_t stream) {
  TTK_REQUIRE(gfeat && y && bn && g, "bc_avgpool_bwd: null pointer");
  TTK_REQUIRE(B > 0 && HW > 0 && C >= 64 && C <= 2048 && C % 64 == 0, "bc_avgpool_bwd: unsupported shape");
  const int slots = 256 / (C / 8);
  hipLaunchKernelGGL(bc_avgpool_bwd_k, dim3(pool_bwd_grid((int64_t)B * HW, C)), dim3(256), (size_t)slots * 2 * C * sizeof(float), (hipStream_t)stream, gfeat,
                     (const bf16_t*)y, bn, (const bf16_t*)skip, (bf16_t*)g, part, B, HW, C);
  TTK_LAUNCH_CHECK("bc_avgpool_bwd");
}

}  // extern "C"
