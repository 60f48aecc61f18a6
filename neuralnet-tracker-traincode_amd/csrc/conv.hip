// Dense convolutions of the ResNet18 backbone variant (reference: backbones/resnet.py:52-104; the arithmetic is
// torchvision.models.resnet.BasicBlock / conv3x3 / conv1x1, un-vendored in the reference) as implicit GEMMs on the
// producer/consumer split kernels of pwconv_f16.hip (fp16 pipe, two pieces per operand, three products; the default)
// or pwconv_split.hip (TTK_GEMM=bf16x3: three bf16 pieces, six products): channels-last activations [B][H][W][C], GEMM
// rows = pixels, contraction = (tap, channel).  Activations are materialised here (post-BatchNorm/ReLU tensors) - ResNet18 is
// matrix-bound (150 flop/B), the extra elementwise passes are ~10 % of its step.
#include "ttk_common.h"
#include "conv_geom.h"

namespace ttk {

// w[Cout][Cin][T] (torch layout, T = KH*KW) -> wf[3][T][Cout][Cin] (forward B operand) and wb[3][T][Cin][Cout] (data
// gradient): bf16 piece planes h, m, l of the exact 3-way split (pwconv_split.hip), so that the GEMM producers move the
// weight operand without arithmetic.
__global__ void conv_weight_repack_k(const float* __restrict__ w, uint16_t* __restrict__ wf, uint16_t* __restrict__ wb, int Cout,
                                     int Cin, int T) {
  const int64_t n = (int64_t)Cout * Cin * T;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T), ci = (int)((i / T) % Cin), co = (int)(i / ((int64_t)T * Cin));
  const float x = w[i];
  const float r1 = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
  const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
  const uint16_t h = (uint16_t)(__float_as_uint(x) >> 16), m = (uint16_t)(__float_as_uint(r1) >> 16), l = (uint16_t)(__float_as_uint(r2) >> 16);
  // planes [K/32][N][32] (pw_split_k's B layout), K = (tap, channel) with the channel fastest
  if (wf) {  // forward: N = Cout, K = (t, ci)
    const size_t o = ((size_t)(t * (Cin >> 5) + (ci >> 5)) * Cout + co) * 32 + (ci & 31);
    wf[o] = h; wf[n + o] = m; wf[2 * n + o] = l;
  }
  if (wb) {  // data gradient: N = Cin, K = (t, co)
    const size_t o = ((size_t)(t * (Cout >> 5) + (co >> 5)) * Cin + ci) * 32 + (co & 31);
    wb[o] = h; wb[n + o] = m; wb[2 * n + o] = l;
  }
}

// fp16 form: max |w| (ordered-uint atomicMax into *wmax, zeroed by the caller), then the two planes of w * pow2_scale(max)
// in the same [K/32][N][32] layouts, each followed by a copy of the maximum (the header the GEMMs read their scale from).
__global__ void __launch_bounds__(256) conv_weight_absmax_k(const float* __restrict__ w, int64_t n, unsigned* __restrict__ wmax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __hip_atomic_load(wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(wmax, __float_as_uint(m));
}

__global__ void conv_weight_repack16_k(const float* __restrict__ w, uint16_t* __restrict__ wf, uint16_t* __restrict__ wb,
                                       const float* __restrict__ wmax, int Cout, int Cin, int T) {
  const int64_t n = (int64_t)Cout * Cin * T;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float mx = *wmax;
  if (i == 0) {
    if (wf && reinterpret_cast<const float*>(wf + 2 * n) != wmax) *reinterpret_cast<float*>(wf + 2 * n) = mx;
    if (wb && reinterpret_cast<const float*>(wb + 2 * n) != wmax) *reinterpret_cast<float*>(wb + 2 * n) = mx;
  }
  const int t = (int)(i % T), ci = (int)((i / T) % Cin), co = (int)(i / ((int64_t)T * Cin));
  const float x = w[i] * pow2_scale(mx);
  const _Float16 hh = (_Float16)x;
  const _Float16 ll = (_Float16)(x - (float)hh);
  const uint16_t h = __builtin_bit_cast(uint16_t, hh), l = __builtin_bit_cast(uint16_t, ll);
  if (wf) {
    const size_t o = ((size_t)(t * (Cin >> 5) + (ci >> 5)) * Cout + co) * 32 + (ci & 31);
    wf[o] = h; wf[n + o] = l;
  }
  if (wb) {
    const size_t o = ((size_t)(t * (Cout >> 5) + (co >> 5)) * Cin + ci) * 32 + (co & 31);
    wb[o] = h; wb[n + o] = l;
  }
}

// ---- all convolutions' weight operands in two launches (ttk_conv_prepare_weights) ----
constexpr int kConvPrepMax = 24;
struct ConvPrepArgs {
  const float* w[kConvPrepMax];
  uint16_t* wf[kConvPrepMax];
  uint16_t* wb[kConvPrepMax];
  int cout[kConvPrepMax], cin[kConvPrepMax], taps[kConvPrepMax];
  int first_chunk[kConvPrepMax + 1];
  int n;
};
__device__ __forceinline__ int conv_prep_layer(const ConvPrepArgs& a) {
  int l = 0;
  while (l + 1 < a.n && (int)blockIdx.x >= a.first_chunk[l + 1]) ++l;
  return l;
}
__device__ __forceinline__ float* conv_prep_hdr(const ConvPrepArgs& a, int l, bool fwd) {
  const int64_t n = (int64_t)a.cout[l] * a.cin[l] * a.taps[l];
  return reinterpret_cast<float*>((fwd ? a.wf[l] : a.wb[l]) + 2 * n);
}
// |w| maxima without atomics: kConvParts workgroups per tensor leave their maxima behind the header (the buffers hold 3 n
// halves, the fp16 form uses 2 n + the header), every repack workgroup folds the kConvParts values of its tensor.
constexpr int kConvParts = 64;
__device__ __forceinline__ float* conv_prep_parts(const ConvPrepArgs& a, int l) { return conv_prep_hdr(a, l, a.wf[l] != nullptr) + 16; }
__global__ void __launch_bounds__(256) conv_prepare_absmax_k(ConvPrepArgs a) {
  __shared__ float sm[4];
  const int l = blockIdx.y;
  const int64_t n4 = (int64_t)a.cout[l] * a.cin[l] * a.taps[l] / 4;  // cout, cin multiples of 32
  const float4* w4 = reinterpret_cast<const float4*>(a.w[l]);
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)kConvParts * 256) {
    const float4 v = w4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) conv_prep_parts(a, l)[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
// One workgroup = one 32 x 32 (co, ci) tile of a tensor over all its taps: the source rows (32 ci x T taps contiguous per co)
// are read coalesced into LDS, and both operand layouts leave as 8-byte pieces of contiguous 2 KB runs
//   wf[t][ci/32][co][ci%32]: the tile's 32 co rows of 32 ci     wb[t][co/32][ci][co%32]: its 32 ci rows of 32 co
// (the element-wise form wrote 2-byte pieces scattered over the planes: 256 us for ResNet18's 11 M weights).
__global__ void __launch_bounds__(256) conv_prepare_repack_k(ConvPrepArgs a) {
  __shared__ float tile[9][32][33];
  const int l = conv_prep_layer(a);
  const int Cout = a.cout[l], Cin = a.cin[l], T = a.taps[l];
  const int64_t n = (int64_t)Cout * Cin * T;
  const int tl = blockIdx.x - a.first_chunk[l], tci = Cin >> 5, co0 = (tl / tci) * 32, ci0 = (tl % tci) * 32;
  uint16_t *wf = a.wf[l], *wb = a.wb[l];
  static_assert(kConvParts == 64, "one value per lane");
  float mx = conv_prep_parts(a, l)[threadIdx.x & 63];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if (tl == 0 && threadIdx.x == 0) {
    if (wf) *conv_prep_hdr(a, l, true) = mx;
    if (wb) *conv_prep_hdr(a, l, false) = mx;
  }
  const float sc = pow2_scale(mx);
  const int rowlen = 32 * T;  // contiguous source floats per co
  for (int e = threadIdx.x; e < 32 * rowlen; e += 256) {
    const int co = e / rowlen, r = e - co * rowlen, ci = r / T, t = r - ci * T;
    tile[t][co][ci] = a.w[l][((int64_t)(co0 + co) * Cin + ci0) * T + r] * sc;
  }
  __syncthreads();
  const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  for (int t = 0; t < T; ++t) {
    if (wf) {  // row = co, 4 consecutive ci
      uint16_t h[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = tile[t][row][c4 + j];
        const _Float16 hh = (_Float16)x;
        h[j] = __builtin_bit_cast(uint16_t, hh);
        lo[j] = __builtin_bit_cast(uint16_t, (_Float16)(x - (float)hh));
      }
      const size_t o = ((size_t)(t * tci + (ci0 >> 5)) * Cout + co0 + row) * 32 + c4;
      *reinterpret_cast<uint2*>(wf + o) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
      *reinterpret_cast<uint2*>(wf + n + o) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
    }
    if (wb) {  // row = ci, 4 consecutive co
      uint16_t h[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = tile[t][c4 + j][row];
        const _Float16 hh = (_Float16)x;
        h[j] = __builtin_bit_cast(uint16_t, hh);
        lo[j] = __builtin_bit_cast(uint16_t, (_Float16)(x - (float)hh));
      }
      const size_t o = ((size_t)(t * (Cout >> 5) + (co0 >> 5)) * Cin + ci0 + row) * 32 + c4;
      *reinterpret_cast<uint2*>(wb + o) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
      *reinterpret_cast<uint2*>(wb + n + o) = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
    }
  }
}

static bool conv_shape_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  return B > 0 && H > 0 && W > 0 && Cin >= 32 && Cin % 32 == 0 && Cout >= 64 && Cout % 64 == 0 && KH == KW && (KH == 1 || KH == 3) &&
         (stride == 1 || stride == 2) && pad == KH / 2;
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_conv_weight_repack(const float* w, void* w_fwd, void* w_bwd, int Cout, int Cin, int KH, int KW, ttk_stream_t stream) {
  TTK_REQUIRE(w && (w_fwd || w_bwd) && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "conv_weight_repack: bad arguments");
  const int64_t n = (int64_t)Cout * Cin * KH * KW;
  if (gemm_mode() == GEMM_F16X2) {
    float* hdr = reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(w_fwd ? w_fwd : w_bwd) + 2 * n);  // behind the two planes
    (void)hipMemsetAsync(hdr, 0, sizeof(float), (hipStream_t)stream);
    hipLaunchKernelGGL(conv_weight_absmax_k, dim3((unsigned)(n / 1024 < 1 ? 1 : (n / 1024 > 256 ? 256 : n / 1024))), dim3(256), 0,
                       (hipStream_t)stream, w, n, reinterpret_cast<unsigned*>(hdr));
    hipLaunchKernelGGL(conv_weight_repack16_k, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w_fwd,
                       (uint16_t*)w_bwd, hdr, Cout, Cin, KH * KW);
    TTK_LAUNCH_CHECK("conv_weight_repack");
  }
  hipLaunchKernelGGL(conv_weight_repack_k, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w_fwd, (uint16_t*)w_bwd,
                     Cout, Cin, KH * KW);
  TTK_LAUNCH_CHECK("conv_weight_repack");
}

int ttk_conv_prepare_weights(int n, const float* const* w, void* const* w_fwd, void* const* w_bwd, const int* cout, const int* cin,
                             const int* ksize, ttk_stream_t stream) {
  TTK_REQUIRE(n > 0 && n <= kConvPrepMax && w && w_fwd && w_bwd && cout && cin && ksize, "conv_prepare_weights: bad arguments (at most %d tensors)", kConvPrepMax);
  for (int i = 0; i < n; ++i)
    TTK_REQUIRE(w[i] && (w_fwd[i] || w_bwd[i]) && cout[i] > 0 && cin[i] > 0 && cout[i] % 32 == 0 && cin[i] % 32 == 0 && (ksize[i] == 1 || ksize[i] == 3),
                "conv_prepare_weights: bad tensor %d", i);
  if (gemm_mode() != GEMM_F16X2) {  // the bf16 comparison path: tensor by tensor
    for (int i = 0; i < n; ++i) {
      const int rc = ttk_conv_weight_repack(w[i], w_fwd[i], w_bwd[i], cout[i], cin[i], ksize[i], ksize[i], stream);
      if (rc) return rc;
    }
    return 0;
  }
  ConvPrepArgs a;
  a.n = n;
  int chunks = 0;
  for (int i = 0; i < n; ++i) {
    a.w[i] = w[i]; a.wf[i] = (uint16_t*)w_fwd[i]; a.wb[i] = (uint16_t*)w_bwd[i];
    a.cout[i] = cout[i]; a.cin[i] = cin[i]; a.taps[i] = ksize[i] * ksize[i];
    a.first_chunk[i] = chunks;
    chunks += (cout[i] / 32) * (cin[i] / 32);  // one workgroup per 32 x 32 (co, ci) tile
  }
  a.first_chunk[n] = chunks;
  hipLaunchKernelGGL(conv_prepare_absmax_k, dim3(kConvParts, n), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(conv_prepare_repack_k, dim3(chunks), dim3(256), 0, (hipStream_t)stream, a);
  TTK_LAUNCH_CHECK("conv_prepare_weights");
}

int ttk_conv_fwd(const float* a_in, const float* a_bound, const void* w_fwd, float* y, float* part, const float* pivot, int B, int H, int W,
                 int Cin, int Cout, int KH, int KW, int stride, int pad, ttk_stream_t stream) {
  TTK_REQUIRE(a_in && a_bound && w_fwd && y, "conv_fwd: null pointer");
  TTK_REQUIRE(conv_shape_ok(B, H, W, Cin, Cout, KH, KW, stride, pad), "conv_fwd: unsupported shape B=%d H=%d W=%d Cin=%d Cout=%d k=%d s=%d p=%d", B, H, W, Cin, Cout, KH, stride, pad);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const ConvGeom geo{H, W, Ho, Wo, stride, pad, KW, Cin, 0};
  const uint16_t* wq = (const uint16_t*)w_fwd;
  const int K = KH * KW * Cin;
  const bool ok = gemm_mode() == GEMM_F16X2
                      ? launch_conv_gemm16(AMODE_PLAIN, EMODE_STATS, a_in, nullptr, nullptr, a_bound, wq, (const float*)(wq + 2 * (size_t)K * Cout), y,
                                           nullptr, const_cast<float*>(pivot), part, (int64_t)B * Ho * Wo, K, Cout, geo, (hipStream_t)stream)
                      : launch_conv_gemm(AMODE_PLAIN, EMODE_STATS, a_in, nullptr, nullptr, wq, y, nullptr, pivot, part, (int64_t)B * Ho * Wo, K,
                                         Cout, geo, (hipStream_t)stream);
  TTK_REQUIRE(ok, "conv_fwd: no kernel for this shape");
  TTK_LAUNCH_CHECK("conv_fwd");
}

// Gradient w.r.t. the conv input.  dy = ga*(g-gmean)+gb*(y-mean) of the conv OUTPUT is formed while loading (bn = that
// output's BatchNorm block).  mask_y/mask_bn (nullable together): the input activation was relu(bn_in(mask_y)) - the
// result is masked with it and part gets the BatchNorm-backward sums (sum g, sum g*(mask_y-mean)) of bn_in; without
// them the raw gradient is written and part is not touched.
int ttk_conv_bwd_data(const float* g, const float* y, const float* bn, const void* w_bwd, const float* mask_y,
                      float* mask_bn, float* g_in, float* part, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                      int stride, int pad, ttk_stream_t stream) {
  TTK_REQUIRE(g && bn && w_bwd && g_in, "conv_bwd_data: null pointer");
  TTK_REQUIRE(y || gemm_mode() == GEMM_F16X2, "conv_bwd_data: a materialised dy (y == NULL) needs the fp16 kernels");
  TTK_REQUIRE((mask_y == nullptr) == (mask_bn == nullptr), "conv_bwd_data: mask_y and mask_bn go together");
  TTK_REQUIRE(conv_shape_ok(B, H, W, Cin, Cout, KH, KW, stride, pad) && Cin % 64 == 0, "conv_bwd_data: unsupported shape");
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const ConvGeom geo{Ho, Wo, H, W, stride, pad, KW, Cout, 1};
  const uint16_t* wq = (const uint16_t*)w_bwd;
  const int K = KH * KW * Cout, em = mask_y ? EMODE_MASK : EMODE_PLAIN;
  const bool ok = gemm_mode() == GEMM_F16X2
                      ? launch_conv_gemm16(y ? AMODE_BNGRAD : AMODE_PLANES, em, g,
                                           y ? y : reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(g) + (size_t)B * Ho * Wo * Cout), bn,
                                           bn + (size_t)TTK_BN_AUX * Cout + TTK_AUX_DY_BOUND, wq,
                                           (const float*)(wq + 2 * (size_t)K * Cin), g_in, mask_y, mask_bn, mask_y ? part : nullptr,
                                           (int64_t)B * H * W, K, Cin, geo, (hipStream_t)stream)
                      : launch_conv_gemm(AMODE_BNGRAD, em, g, y, bn, wq, g_in, mask_y, mask_bn, mask_y ? part : nullptr, (int64_t)B * H * W, K, Cin, geo,
                                         (hipStream_t)stream);
  TTK_REQUIRE(ok, "conv_bwd_data: no kernel for this shape");
  TTK_LAUNCH_CHECK("conv_bwd_data");
}

// dw[Cout][Cin][KH][KW] += sum_{pixels} dy (x) a_in.  The caller zeroes dw (or accumulates on purpose).
size_t ttk_conv_wgrad_partial_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  if (gemm_mode() != GEMM_F16X2 || !conv_shape_ok(B, H, W, Cin, Cout, KH, KW, stride, pad)) return 0;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  return conv_wgrad16_partial_bytes((int64_t)B * Ho * Wo, Cout, KH * KW * Cin, KH * KW);
}

int ttk_conv_bwd_weight(const float* g, const float* y, const float* bn, const float* a_in, const float* a_bound, float* dw, float* partial,
                        int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, ttk_stream_t stream) {
  TTK_REQUIRE(g && bn && a_in && a_bound && dw, "conv_bwd_weight: null pointer");
  TTK_REQUIRE(y || gemm_mode() == GEMM_F16X2, "conv_bwd_weight: a materialised dy (y == NULL) needs the fp16 kernels");
  TTK_REQUIRE(conv_shape_ok(B, H, W, Cin, Cout, KH, KW, stride, pad), "conv_bwd_weight: unsupported shape");
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const ConvGeom geo{H, W, Ho, Wo, stride, pad, KW, Cin, 0};
  const bool ok = gemm_mode() == GEMM_F16X2
                      ? launch_conv_wgrad16(g, y, bn, a_in, a_bound, dw, partial, (int64_t)B * Ho * Wo, Cout, KH * KW, geo, (hipStream_t)stream)
                      : launch_conv_wgrad(g, y, bn, a_in, dw, (int64_t)B * Ho * Wo, Cout, KH * KW, geo, (hipStream_t)stream);
  TTK_REQUIRE(ok, "conv_bwd_weight: no kernel for this shape");
  TTK_LAUNCH_CHECK("conv_bwd_weight");
}

}  // extern "C"
