// bn_act: materialise a = max(scale*(y - mean) + beta (+ skip), 0) of one layer as a plain channels-last tensor - what
// MobileNet.forward hands out as its intermediate feature maps (reference backbones/mobilenet_v1.py:165-186).  The depthwise
// convolution itself (forward, data gradient with the fused weight gradient) lives in dwconv_tiled.hip.
#include "ttk_common.h"

namespace ttk {

// y, skip: channel blocks [C/32][rows][32] (ttk_common.h act_off); a: [rows][C].  Thread = (row, channel quad).
__global__ void __launch_bounds__(kBlock) bn_act_k(const float* __restrict__ y, const float* __restrict__ bnp,
                                                    const float* __restrict__ skip,
                                                    float* __restrict__ a, int64_t rows, int C, int qshift) {
  const int quads = C >> 2;
  const int c4 = threadIdx.x & (quads - 1);
  const BnApply4 bn = BnApply4::load(bnp, C, 4 * c4);
  const int64_t items = rows << qshift;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    const size_t off = act_off(idx >> qshift, 4 * c4, rows);
    st4(a + ((size_t)idx << 2), skip ? bn.act(ld4(y + off), ld4(skip + off)) : bn.act(ld4(y + off)));
  }
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_bn_act(const float* y, const float* bn, const float* skip, float* a, int64_t rows, int C, ttk_stream_t stream) {
  TTK_REQUIRE(y && bn && a, "bn_act: null pointer");
  TTK_REQUIRE(rows > 0 && C >= 32 && C <= 1024 && (C & (C - 1)) == 0, "bn_act: unsupported shape rows=%lld C=%d", (long long)rows, C);
  const int64_t items = rows * (C / 4);
  int g = (int)ceil_div(items, kBlock);
  if (g > 4096) g = 4096;
  int qs = 0;
  while ((1 << qs) < C / 4) ++qs;
  hipLaunchKernelGGL(bn_act_k, dim3(g), dim3(kBlock), 0, (hipStream_t)stream, y, bn, skip, a, rows, C, qs);
  TTK_LAUNCH_CHECK("bn_act");
}

}  // extern "C"
