// Global average pooling over the last block's output (AdaptiveAvgPool2d(1) + view,
// backbones/mobilenet_v1.py:143,180-181) with the last BatchNorm+residual+ReLU applied on load, and
// its backward, which starts the chain of BatchNorm-backward partial sums.
#include "ttk_common.h"

namespace ttk {

// element offset of (pixel m, channel c) by the layout argument: 0 = channel blocks of 32 (the fp32 MobileNet path), 1 = channels-last rows
// (ResNet18), 2 = channel blocks of 64 (the bf16-compute path, csrc/bc_common.h; C >= 64)
__device__ __forceinline__ size_t pool_off(int layout, int64_t m, int c, int64_t M, int C) {
  if (layout == 1) return (size_t)m * C + c;
  if (layout == 2) return ((size_t)(c >> 6) * (size_t)M + (size_t)m) * 64 + (c & 63);
  return act_off(m, c, M);
}

// thread = (sample, channel quad)
template <typename T>
__global__ void __launch_bounds__(kBlock) avgpool_fwd_k(const T* __restrict__ y, const float* __restrict__ bnp,
                                                         const T* __restrict__ skip,
                                                         float* __restrict__ feat, int B, int HW, int C, int rows_layout) {
  const int quads = C >> 2;
  const int64_t items = (int64_t)B * quads;
  const float inv = 1.0f / (float)HW;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    const int c4 = (int)(idx % quads);
    const int n = (int)(idx / quads);
    const BnApply4 bn = BnApply4::load(bnp, C, 4 * c4);
    float4 s = f4(0.f);
    for (int p = 0; p < HW; ++p) {
      const int64_t m = (int64_t)n * HW + p;
      const size_t off = pool_off(rows_layout, m, 4 * c4, (int64_t)B * HW, C);
      s = add4(s, skip ? bn.act(Act<T>::ld(y + off), Act<T>::ld(skip + off)) : bn.act(Act<T>::ld(y + off)));
    }
    st4(feat + (size_t)n * C + 4 * c4, make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv));
  }
}

// thread = (sample, pixel, channel quad)
template <typename T, typename TG>
__global__ void __launch_bounds__(kBlock) avgpool_bwd_k(const float* __restrict__ gfeat, const T* __restrict__ y,
                                                         float* __restrict__ bnp, const T* __restrict__ skip, TG* __restrict__ g,
                                                         float* __restrict__ part, int B, int HW, int C, int qshift, int rows_layout) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int quads = C >> 2;
  const int c4 = threadIdx.x & (quads - 1);
  const BnApply4 bn = BnApply4::load(bnp, C, 4 * c4);
  const float inv = 1.0f / (float)HW;
  const int64_t items = ((int64_t)B * HW) << qshift;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  float gmx = 0.f;  // max |g| (ttk.h, TTK_AUX_GMAX)
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    const int n = (int)((unsigned)(idx >> qshift) / (unsigned)HW);  // 32-bit division (the host checks that B * HW fits)
    const size_t off = pool_off(rows_layout, idx >> qshift, 4 * c4, (int64_t)B * HW, C);
    const float4 yv = Act<T>::ld(y + off);
    const float4 a = skip ? bn.act(yv, Act<T>::ld(skip + off)) : bn.act(yv);
    float4 gv = ld4(gfeat + (size_t)n * C + 4 * c4);
    gv = Act<TG>::round(mask4(make_float4(gv.x * inv, gv.y * inv, gv.z * inv, gv.w * inv), a));  // sums / maximum of what is stored
    Act<TG>::st(g + off, gv);
    gmx = fmaxf(fmaxf(gmx, fmaxf(fabsf(gv.x), fabsf(gv.y))), fmaxf(fabsf(gv.z), fabsf(gv.w)));
    s1 = add4(s1, gv);
    s2 = fma4(gv, sub4(yv, bn.mean), s2);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) gmx = fmaxf(gmx, __shfl_xor(gmx, off));
  if ((threadIdx.x & 63) == 0) {  // most waves find the slot already at or above their maximum: one relaxed read instead of ~3000 atomics on one address
    unsigned* slot = reinterpret_cast<unsigned*>(bnp + (size_t)TTK_BN_AUX * C + TTK_AUX_GMAX);
    if (__float_as_uint(gmx) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, __float_as_uint(gmx));
  }
  if (part) block_channel_partials<1024>(s1, s2, c4, C, part + (size_t)blockIdx.x * 2 * C, smem);
}

}  // namespace ttk

using namespace ttk;

static int log2i_(int v) {
  int r = 0;
  while ((1 << r) < v) ++r;
  return r;
}

extern "C" {

int ttk_avgpool_fwd(const void* y, const float* bn, const void* skip, float* feat, int B, int HW,
                    int C, int act_bf16, ttk_stream_t stream) {
  TTK_REQUIRE(y && bn && feat, "avgpool_fwd: null pointer");
  TTK_REQUIRE(B > 0 && HW > 0 && C >= 32 && C <= 1024 && (C & (C - 1)) == 0, "avgpool_fwd: unsupported shape B=%d HW=%d C=%d", B, HW, C);
  const int64_t items = (int64_t)B * (C / 4);
  TTK_ACT_DISPATCH(act_bf16, hipLaunchKernelGGL((avgpool_fwd_k<ActT>), dim3(elementwise_grid(items)), dim3(kBlock), 0, (hipStream_t)stream,
                                                (const ActT*)y, bn, (const ActT*)skip, feat, B, HW, C, (act_bf16 & TTK_LAYOUT_ROWS) ? 1 : ((act_bf16 & TTK_LAYOUT_CB64) ? 2 : 0)));
  TTK_LAUNCH_CHECK("avgpool_fwd");
}

int ttk_avgpool_bwd(const float* gfeat, const void* y, float* bn, const void* skip, void* g,
                    float* part, int B, int HW, int C, int act_bf16, ttk_stream_t stream) {
  TTK_REQUIRE(gfeat && y && bn && g, "avgpool_bwd: null pointer");
  TTK_REQUIRE(B > 0 && HW > 0 && C >= 32 && C <= 1024 && (C & (C - 1)) == 0, "avgpool_bwd: unsupported shape");
  TTK_REQUIRE((int64_t)B * HW < (int64_t)1 << 31, "avgpool_bwd: too many pixels for 32-bit indexing");
  const int qs = log2i_(C / 4);
  const int64_t items = ((int64_t)B * HW) << qs;
  TTK_ACT_DISPATCH(act_bf16, hipLaunchKernelGGL((avgpool_bwd_k<ActT, GradT>), dim3(elementwise_grid(items)), dim3(kBlock), 2 * (size_t)C * sizeof(float),
                                                (hipStream_t)stream, gfeat, (const ActT*)y, bn, (const ActT*)skip, (GradT*)g, part, B, HW, C, qs, (act_bf16 & TTK_LAYOUT_ROWS) ? 1 : ((act_bf16 & TTK_LAYOUT_CB64) ? 2 : 0)));
  TTK_LAUNCH_CHECK("avgpool_bwd");
}

}  // extern "C"
