// The non-GEMM kernels of the ResNet18 backbone variant (reference: backbones/resnet.py:52-104 = torchvision ResNet
// with a 1-channel 7x7 stem; BasicBlock arithmetic restated from torchvision.models.resnet, which the reference does
// not vendor): stem convolution and its weight gradient, 3x3/s2 max-pool with BatchNorm+ReLU on load, the residual
// add + ReLU that materialises a block's output activation, and the elementwise backward of that add.  All HBM-bound;
// the dense 3x3 / 1x1 convolutions are in conv.hip.
#include "ttk_common.h"
#include "stem_wgrad.h"
#include <stdlib.h>

namespace ttk {

constexpr int kS7C = 64;      // stem output channels
constexpr int kS7K = 7;       // 7x7, stride 2, pad 3
constexpr int kS7Half = 32;   // channels per workgroup column (blockIdx.y)
constexpr int kS7Wp = 192;    // stem7_fwd_mfma_k: row pitch of a wave's input patch = three 64-lane strips

// thread = TWO horizontally adjacent output pixels x 32 channels (same scheme as stem.hip, see there): the filter
// bank of this channel half sits in LDS as wt[tap][c] and is read with wave-uniform (broadcast) ds_read_b128.
__global__ void __launch_bounds__(kBlock) stem7_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, float* __restrict__ part, const float* __restrict__ pivot,
                                                       int B, int H, int W, int Ho, int Wo) {
  __shared__ __attribute__((aligned(16))) float wt[kS7K * kS7K][kS7Half];
  __shared__ float red[kBlock / kWave][2 * kS7Half];
  const int cbase = blockIdx.y * kS7Half;
  for (int i = threadIdx.x; i < kS7K * kS7K * kS7Half; i += kBlock) {
    const int t = i % (kS7K * kS7K), c = i / (kS7K * kS7K);
    wt[t][c] = w[(size_t)(cbase + c) * kS7K * kS7K + t];  // w[c][tap] -> wt[tap][c]
  }
  __syncthreads();
  float s1[kS7Half], s2[kS7Half];
#pragma unroll
  for (int c = 0; c < kS7Half; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  const int Wpairs = (Wo + 1) / 2;
  const int64_t npairs = (int64_t)B * Ho * Wpairs;
  for (int64_t pr = (int64_t)blockIdx.x * kBlock + threadIdx.x; pr < npairs; pr += (int64_t)gridDim.x * kBlock) {
    asm volatile("" ::: "memory");  // keep the LDS filter reads inside the iteration (see stem.hip)
    const int wp = (int)(pr % Wpairs), ho = (int)((pr / Wpairs) % Ho), n = (int)(pr / ((int64_t)Wpairs * Ho));
    const int wo = 2 * wp;
    const bool second = wo + 1 < Wo;
    const float* xn = x + (size_t)n * H * W;
    float xin[kS7K][kS7K + 2];
#pragma unroll
    for (int kh = 0; kh < kS7K; ++kh) {
      const int hi = 2 * ho + kh - 3;
#pragma unroll
      for (int k = 0; k < kS7K + 2; ++k) {
        const int wi = 2 * wo + k - 3;
        xin[kh][k] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? xn[(size_t)hi * W + wi] : 0.f;
      }
    }
    float* y0 = y + (((size_t)n * Ho + ho) * Wo + wo) * kS7C + cbase;
#pragma unroll
    for (int c4 = 0; c4 < kS7Half / 4; ++c4) {
      float4 a0 = f4(0.f), a1 = f4(0.f);
#pragma unroll
      for (int kh = 0; kh < kS7K; ++kh)
#pragma unroll
        for (int kw = 0; kw < kS7K; ++kw) {
          const float4 wq = ld4(&wt[kh * kS7K + kw][4 * c4]);
          a0 = fma4(f4(xin[kh][kw]), wq, a0);
          a1 = fma4(f4(xin[kh][kw + 2]), wq, a1);
        }
      st4(y0 + 4 * c4, a0);
      const float4 pv = pivot ? ld4(pivot + cbase + 4 * c4) : f4(0.f);  // the sums are those of y - pivot
      a0 = sub4(a0, pv);
      if (second) { st4(y0 + kS7C + 4 * c4, a1); a1 = sub4(a1, pv); }
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
    for (int c = 0; c < kS7Half; ++c) {
      float a = s1[c], b = s2[c];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
      if (lane == 0) { red[wv][c] = a; red[wv][kS7Half + c] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kS7Half) {
      const int which = threadIdx.x / kS7Half, c = threadIdx.x % kS7Half;
      float a = 0.f;
      for (int i = 0; i < kBlock / kWave; ++i) a += red[i][threadIdx.x];  // fixed wave order
      part[(size_t)blockIdx.x * 2 * kS7C + (size_t)which * kS7C + cbase + c] = a;
    }
  }
}

// The same convolution on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation - an fma
// chain's accuracy): 13.6 GFLOP at B = 512 ran VALU-bound (0.53 ms) on the kernel above; the output's 92 us of HBM writes
// are the floor.  One WAVE = 32 consecutive output pixels (row-major inside an image) x 64 channels: the (at most 9)
// input rows its pixels touch are staged in a private LDS patch [9][Wp] (zero padded), MFMA j multiplies the pixels'
// values under taps 2j, 2j+1 (lanes 0-31 | 32-63: one ds_read_b32 each, conflict-free at stride 2) with the two filter
// rows held in registers.  Wave-private LDS: no workgroup barrier in the loop; groups go round-robin over all waves.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(kBlock) stem7_fwd_mfma_k(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y, float* __restrict__ part, const float* __restrict__ pivot,
                                                            int B, int H, int W, int Ho, int Wo, int PR, int wave_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem7[];
  __shared__ float red[kBlock / kWave][2 * kS7C];
  constexpr int kTaps = kS7K * kS7K, kSteps = (kTaps + 1) / 2;  // 49 taps, 25 MFMAs (the 50th tap has zero weights)
  constexpr int kLdo = kS7C + 4;                                 // row pitch of the output image in LDS
  constexpr int Wp = kS7Wp;                                      // row pitch of the input patch (3 zero columns left; W + 7 <= Wp)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, px = lane & 31, half = lane >> 5;
  float* patch = smem7 + (size_t)wv * wave_floats;  // [PR][Wp] input rows, then (aliased) the [32][kLdo] output image
  // B operand: lane (n = px, k = half) of MFMA j and channel tile t holds w[32 t + px][2 j + half]
  float wreg[2][kSteps];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < kSteps; ++j) {
      const int tap = 2 * j + half;
      wreg[t][j] = tap < kTaps ? w[(size_t)(32 * t + px) * kTaps + tap] : 0.f;
    }
  // LDS offset of the tap of MFMA j relative to the pixel's window origin, for the lower / upper lane half
  auto toff = [](int j, int hf) {
    const int tap = 2 * j + hf < kTaps ? 2 * j + hf : 0, kh = tap / kS7K;
    return kh * Wp + tap - kh * kS7K;
  };
  const int hw = Ho * Wo, gpi = (hw + 31) / 32;  // 32-pixel groups per image
  const int groups = B * gpi, nwaves = (int)gridDim.x * (kBlock / kWave);  // (the host checks B * gpi < 2^31)
  const int c4 = lane & 15, prow = lane >> 4;  // output pass: 16 lanes x float4 = one pixel's 64 channels, 4 pixels per instruction
  const float4 pv = pivot ? ld4(pivot + 4 * c4) : f4(0.f);  // the sums are those of y - pivot
  float4 s1 = f4(0.f), s2 = f4(0.f);
  // The input rows of a group are fetched into registers one group ahead (27 independent loads in flight under the
  // previous group's MFMAs and output pass; a load -> LDS-write loop paid one memory latency per element, 13 us per group).
  // Nine rows x three 64-column strips cover every patch of up to 9 rows x 192 columns; taller patches (narrow images:
  // 32 pixels span several rows) take further rounds without the overlap.
  float v[9][3];
  auto fetch = [&](int g, int r0) {
    const int n = g / gpi, p0 = (g - n * gpi) * 32, oh0 = p0 / Wo;
    const float* xn = x + (size_t)n * H * W;
#pragma unroll
    for (int rr = 0; rr < 9; ++rr) {
      const int hi = 2 * oh0 - 3 + r0 + rr;
      const bool rok = r0 + rr < PR && hi >= 0 && hi < H;
      const float* xr = xn + (size_t)(rok ? hi : 0) * W;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int wi = lane + 64 * k - 3;
        const bool ok = rok && wi >= 0 && wi < W;
        const float t = xr[ok ? wi : 0];  // unconditional (clamped) load: a predicated one is a branch with its own wait
        v[rr][k] = ok ? t : 0.f;
      }
    }
  };
  auto stash = [&](int r0) {
#pragma unroll
    for (int rr = 0; rr < 9; ++rr)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        patch[(r0 + rr) * Wp + lane + 64 * k] = v[rr][k];  // unconditional: the patch holds whole rounds of 9 rows x 192 columns
      }
  };
  const int g_first = (int)blockIdx.x * (kBlock / kWave) + wv;
  if (g_first < groups) fetch(g_first, 0);
  for (int gidx = g_first; gidx < groups; gidx += nwaves) {
    const int n = gidx / gpi, p0 = (gidx - n * gpi) * 32;
    const int oh0 = p0 / Wo;
    __builtin_amdgcn_wave_barrier();  // (same wave, LDS in order) the previous group's output pass is done before the patch is overwritten
    stash(0);
    for (int r0 = 9; r0 < PR; r0 += 9) { fetch(gidx, r0); stash(r0); }
    if (gidx + nwaves < groups) fetch(gidx + nwaves, 0);
    __builtin_amdgcn_wave_barrier();
    const int p = p0 + px;
    const int pc = p < hw ? p : hw - 1, oh = pc / Wo, ow = pc - oh * Wo;
    const float* win = patch + (2 * (oh - oh0)) * Wp + 2 * ow;
    f32x16_t acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < kSteps; ++j) {
      const float a = win[half ? toff(j, 1) : toff(j, 0)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wreg[0][j], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wreg[1][j], acc1, 0, 0, 0);
    }
    // accumulator element e of lane (n = px, half): pixel row (e & 3) + 8 (e >> 2) + 4 half of the group, channel px (+32).
    // Through LDS (the patch is dead) so that the stores are 16 bytes per lane, whole 256-byte pixel rows
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      patch[row * kLdo + px] = acc0[e];
      patch[row * kLdo + 32 + px] = acc1[e];
    }
    __builtin_amdgcn_wave_barrier();
    float* yg = y + ((size_t)n * hw + p0) * kS7C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = prow + 4 * i;
      if (p0 + row < hw) {
        float4 v = ld4(patch + row * kLdo + 4 * c4);
        st4(yg + (size_t)row * kS7C + 4 * c4, v);
        v = sub4(v, pv);
        s1 = add4(s1, v);
        s2 = fma4(v, v, s2);
      }
    }
  }
  if (part) {  // channels 4 c4 .. 4 c4 + 3: fold the four lanes of a quad column, then the waves in a fixed order
    s1.x += __shfl_xor(s1.x, 16); s1.y += __shfl_xor(s1.y, 16); s1.z += __shfl_xor(s1.z, 16); s1.w += __shfl_xor(s1.w, 16);
    s2.x += __shfl_xor(s2.x, 16); s2.y += __shfl_xor(s2.y, 16); s2.z += __shfl_xor(s2.z, 16); s2.w += __shfl_xor(s2.w, 16);
    s1.x += __shfl_xor(s1.x, 32); s1.y += __shfl_xor(s1.y, 32); s1.z += __shfl_xor(s1.z, 32); s1.w += __shfl_xor(s1.w, 32);
    s2.x += __shfl_xor(s2.x, 32); s2.y += __shfl_xor(s2.y, 32); s2.z += __shfl_xor(s2.z, 32); s2.w += __shfl_xor(s2.w, 32);
    if (lane < 16) { st4(&red[wv][4 * c4], s1); st4(&red[wv][kS7C + 4 * c4], s2); }
    __syncthreads();
    if (threadIdx.x < 2 * kS7C) {
      float a = 0.f;
      for (int i = 0; i < kBlock / kWave; ++i) a += red[i][threadIdx.x];
      part[(size_t)blockIdx.x * 2 * kS7C + threadIdx.x] = a;
    }
  }
}

// ---- max-pool 3x3 / stride 2 / pad 1 over relu(bn(y)); idx = window position (kh*3+kw) of the FIRST maximum, which is
// what torch's max_pool2d backward routes the gradient to.  thread = (output pixel, channel quad).
__global__ void __launch_bounds__(kBlock) maxpool_fwd_k(const float* __restrict__ y, float* __restrict__ bnp,
                                                         float* __restrict__ a, unsigned char* __restrict__ idx, int B, int H,
                                                         int W, int Ho, int Wo, int C) {
  const int quads = C >> 2;
  const int c4 = threadIdx.x & (quads - 1);
  const BnApply4 bn = BnApply4::load(bnp, C, 4 * c4);
  const unsigned items = (unsigned)B * Ho * Wo * quads, qshift = __builtin_ctz(quads);  // 32-bit index arithmetic (checked by the host)
  float amax = 0.f;
  for (unsigned it = blockIdx.x * kBlock + threadIdx.x; it < items; it += gridDim.x * kBlock) {
    const unsigned pix = it >> qshift, row = pix / (unsigned)Wo;
    const int wo = (int)(pix - row * (unsigned)Wo), n = (int)(row / (unsigned)Ho), ho = (int)(row - (unsigned)n * (unsigned)Ho);
    float4 m = f4(-1.f);  // relu outputs are >= 0 and every window holds at least one pixel
    uchar4 am = make_uchar4(0, 0, 0, 0);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = 2 * ho + kh - 1;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = 2 * wo + kw - 1;
        if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
        const float4 v = bn.act(ld4(y + (((size_t)n * H + hi) * W + wi) * C + 4 * c4));
        const unsigned char t = (unsigned char)(kh * 3 + kw);
        if (v.x > m.x) { m.x = v.x; am.x = t; }
        if (v.y > m.y) { m.y = v.y; am.y = t; }
        if (v.z > m.z) { m.z = v.z; am.z = t; }
        if (v.w > m.w) { m.w = v.w; am.w = t; }
      }
    }
    st4(a + (size_t)it * 4, m);
    *reinterpret_cast<uchar4*>(idx + (size_t)it * 4) = am;
    amax = fmaxf(amax, fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w)));
  }
  wave_raise_max(bnp + (size_t)TTK_BN_AUX * C + TTK_AUX_ACT_BOUND, amax);  // the bound the consuming convolutions scale a by
}

// gradient w.r.t. the BatchNorm output of y (already through the ReLU mask) + that BatchNorm's backward sums.
// ga (+ gb): gradient(s) w.r.t. the pooled activation.  thread = (input pixel, channel quad).
// 1024 threads per workgroup: the partial rows cap this streaming gather at 1024 workgroups, and four waves each left a CU with
// 16 waves in flight (377 -> 304 us at B = 512)
__global__ void __launch_bounds__(1024) maxpool_bwd_k(const float* __restrict__ ga, const float* __restrict__ gb,
                                                         const unsigned char* __restrict__ idx, const float* __restrict__ y,
                                                         const float* __restrict__ bnp, float* __restrict__ g,
                                                         float* __restrict__ part, int B, int H, int W, int Ho, int Wo, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int quads = C >> 2;
  const int c4 = threadIdx.x & (quads - 1);
  const BnApply4 bn = BnApply4::load(bnp, C, 4 * c4);
  // 32-bit index arithmetic (the host checks that the item count fits): the four 64-bit divisions per item cost this gather
  // kernel 9 % (414 -> 378 us at B = 512; two items in flight per thread on top of it: no further gain)
  const unsigned items = (unsigned)B * H * W * quads, qshift = __builtin_ctz(quads);
  float4 s1 = f4(0.f), s2 = f4(0.f);
  for (unsigned it = blockIdx.x * 1024 + threadIdx.x; it < items; it += gridDim.x * 1024) {
    const unsigned pix = it >> qshift, row = pix / (unsigned)W;
    const int wi = (int)(pix - row * (unsigned)W), n = (int)(row / (unsigned)H), hi = (int)(row - (unsigned)n * (unsigned)H);
    const float4 yv = ld4(y + (size_t)it * 4);
    const float4 act = bn.act(yv);
    float4 acc = f4(0.f);
    // windows (ho, wo) that contain (hi, wi): 2*ho - 1 <= hi <= 2*ho + 1
    for (int ho = (hi + 1) >> 1; ho >= (hi >> 1) && ho >= 0; --ho) {
      if (ho >= Ho) continue;
      const int kh = hi - 2 * ho + 1;
      for (int wo = (wi + 1) >> 1; wo >= (wi >> 1) && wo >= 0; --wo) {
        if (wo >= Wo) continue;
        const unsigned char t = (unsigned char)(kh * 3 + (wi - 2 * wo + 1));
        const size_t o = ((((size_t)n * Ho + ho) * Wo + wo) * quads + c4) * 4;
        const uchar4 am = *reinterpret_cast<const uchar4*>(idx + o);
        float4 gv = ld4(ga + o);
        if (gb) gv = add4(gv, ld4(gb + o));
        if (am.x == t) acc.x += gv.x;
        if (am.y == t) acc.y += gv.y;
        if (am.z == t) acc.z += gv.z;
        if (am.w == t) acc.w += gv.w;
      }
    }
    const float4 gv = mask4(acc, act);
    st4(g + (size_t)it * 4, gv);
    s1 = add4(s1, gv);
    s2 = fma4(gv, sub4(yv, bn.mean), s2);
  }
  if (part) block_channel_partials<1024, 1024>(s1, s2, c4, C, part + (size_t)blockIdx.x * 2 * C, smem);
}

// a = relu(bn(y) + r), r = res (an activation) or res_bn(res) (a raw conv output, the downsample branch) or nothing
__global__ void __launch_bounds__(kBlock) bn_add_act_k(const float* __restrict__ y, float* __restrict__ bnp,
                                                        const float* __restrict__ res, const float* __restrict__ res_bn,
                                                        float* __restrict__ a, const float* __restrict__ res_bound, int measure,
                                                        int64_t items, int C) {
  const int quads = C >> 2;
  const int c4 = threadIdx.x & (quads - 1);
  const BnApply4 bn = BnApply4::load(bnp, C, 4 * c4);
  BnApply4 rb = bn;
  if (res_bn) rb = BnApply4::load(res_bn, C, 4 * c4);
  float amax = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    const size_t off = (size_t)idx << 2;
    float4 v = bn.pre(ld4(y + off));
    if (res) v = add4(v, res_bn ? rb.pre(ld4(res + off)) : ld4(res + off));
    v = relu4(v);
    st4(a + off, v);
    if (measure) amax = fmaxf(amax, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
  }
  // The bound the consuming convolutions scale a by.  Training: relu(bn(y) + r) <= bound(relu(bn(y))) + bound(r), both already
  // known from the batch statistics (ttk_bn_fwd_finalize) - measuring the maximum instead cost as much as the pass itself
  // (tens of thousands of waves finishing together and racing for one atomic).  Without statistics (eval): the measured maximum.
  float* slot = bnp + (size_t)TTK_BN_AUX * C + TTK_AUX_ACT_BOUND;
  if (measure) wave_raise_max(slot, amax);
  else if (res_bound && blockIdx.x == 0 && threadIdx.x == 0) *slot += *res_bound;  // nothing else touches the slot during this launch
}

// Weight gradient of the 7x7 stem on the fp32 matrix pipe:  dW[c][tap] = sum_pixels dy[pixel][c] * x[tap of pixel],
// dy = ga*(g-gmean) + gb*(y-mean) formed on load.  The VALU kernel of stem_wgrad.h is compute-bound here (13.6 GFLOP at
// B = 512: 0.49 ms against 0.19 ms of HBM time).  As the forward kernel, one WAVE = 32 consecutive output pixels: their
// input rows sit in a private LDS patch, dy[32][64] beside it; v_mfma_f32_32x32x2_f32 with M = channels (2 tiles), N = taps
// (2 tiles: 49 of 64), K = pixel pairs - 64 MFMAs per group, accumulated in registers over all groups of the wave.  Patch
// pitch 201: the lanes of a B fragment are taps, bank (9 kh + kw) mod 32 - at most 2-way.  The workgroup's sum leaves as a
// row of `partial` (folded in a fixed order by the caller) or, without scratch, as one atomicAdd per weight.
constexpr int kS7WpW = 201;
__global__ void __launch_bounds__(kBlock) stem7_wgrad_mfma_k(const float* __restrict__ g, const float* __restrict__ y,
                                                              const float* __restrict__ bn, const float* __restrict__ x,
                                                              float* __restrict__ dw, float* __restrict__ partial, int B, int H, int W,
                                                              int Ho, int Wo, int PR, int patch_floats, int wave_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem7[];
  constexpr int kTaps = kS7K * kS7K, Wp = kS7WpW, kLd = kS7C + 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n32 = lane & 31, half = lane >> 5;
  float* patch = smem7 + (size_t)wv * wave_floats;  // [rounds of 9][Wp] input rows
  float* dyS = patch + patch_floats;                // [32][kLd]
  // taps of this lane in the two N tiles (the second tile holds taps 32..48)
  const int tapA = n32, tapB = 32 + n32;
  const bool okB = tapB < kTaps;
  const int offA = (tapA / kS7K) * Wp + tapA % kS7K;
  const int offB = okB ? (tapB / kS7K) * Wp + tapB % kS7K : 0;
  const int c4 = lane & 15, prow = lane >> 4;  // dy staging: 16 lanes x float4 = one pixel's 64 channels
  const float4 ga = ld4(bn + TTK_BN_GA * kS7C + 4 * c4), gb = ld4(bn + TTK_BN_GB * kS7C + 4 * c4);
  const float4 gmean = ld4(bn + TTK_BN_GMEAN * kS7C + 4 * c4), mean = ld4(bn + TTK_BN_MEAN * kS7C + 4 * c4);
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int hw = Ho * Wo, gpi = (hw + 31) / 32;
  const int groups = B * gpi, nwaves = (int)gridDim.x * (kBlock / kWave);
  float v[9][3];
  float4 rg[8], ry[8];
  auto fetch_x = [&](int gi, int r0) {
    const int n = gi / gpi, p0 = (gi - n * gpi) * 32, oh0 = p0 / Wo;
    const float* xn = x + (size_t)n * H * W;
#pragma unroll
    for (int rr = 0; rr < 9; ++rr) {
      const int hi = 2 * oh0 - 3 + r0 + rr;
      const bool rok = r0 + rr < PR && hi >= 0 && hi < H;
      const float* xr = xn + (size_t)(rok ? hi : 0) * W;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int wi = lane + 64 * k - 3;
        const bool ok = rok && wi >= 0 && wi < W;
        const float t = xr[ok ? wi : 0];  // unconditional (clamped) load
        v[rr][k] = ok ? t : 0.f;
      }
    }
  };
  auto fetch_gy = [&](int gi) {
    const int n = gi / gpi, p0 = (gi - n * gpi) * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = p0 + prow + 4 * i;
      const size_t o = ((size_t)n * hw + (p < hw ? p : hw - 1)) * kS7C + 4 * c4;
      rg[i] = ld4nt(g + o);
      ry[i] = ld4nt(y + o);
    }
  };
  // Input rows and g / y of a group are fetched one group ahead (in flight under the previous group's MFMAs).  Measured
  // alternatives, all slower: validity applied at the LDS write instead of behind the load (655 us), all loads issued at the
  // group's start and waited for once (663 us) - this compiler serialises the 27 scalar loads in both.
  const int g_first = (int)blockIdx.x * (kBlock / kWave) + wv;
  if (g_first < groups) { fetch_x(g_first, 0); fetch_gy(g_first); }
  for (int gidx = g_first; gidx < groups; gidx += nwaves) {
    const int n = gidx / gpi, p0 = (gidx - n * gpi) * 32, oh0 = p0 / Wo;
    __builtin_amdgcn_wave_barrier();  // the previous group's fragment reads are done (same wave, LDS in order)
#pragma unroll
    for (int rr = 0; rr < 9; ++rr)
#pragma unroll
      for (int k = 0; k < 3; ++k) patch[rr * Wp + lane + 64 * k] = v[rr][k];
    for (int r0 = 9; r0 < PR; r0 += 9) {
      fetch_x(gidx, r0);
#pragma unroll
      for (int rr = 0; rr < 9; ++rr)
#pragma unroll
        for (int k = 0; k < 3; ++k) patch[(r0 + rr) * Wp + lane + 64 * k] = v[rr][k];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pl = prow + 4 * i;
      float4 d = fma4(ga, sub4(rg[i], gmean), mul4(gb, sub4(ry[i], mean)));
      if (p0 + pl >= hw) d = f4(0.f);  // pixels past the image end contribute nothing
      st4(dyS + pl * kLd + 4 * c4, d);
    }
    if (gidx + nwaves < groups) { fetch_x(gidx + nwaves, 0); fetch_gy(gidx + nwaves); }  // in flight under the MFMAs
    __builtin_amdgcn_wave_barrier();
    // pixel of this lane's k slot: p0 + 2 j + half, walked with a branch-free carry; the fragments of pixel pair j + 1 are
    // requested before the four MFMAs of pair j issue (waiting for each ds_read right in front of its MFMA left the matrix
    // pipe idle for an LDS latency per pair)
    int pidx = p0 + half;
    int oh = (pidx < hw ? pidx : hw - 1) / Wo, ow = (pidx < hw ? pidx : hw - 1) - oh * Wo;
    float fa0[2], fa1[2], fbA[2], fbB[2];
    auto frag = [&](int j, int set) {
      const float* win = patch + (2 * (oh - oh0)) * Wp + 2 * ow;
      fbA[set] = win[offA];
      fbB[set] = win[offB];
      fa0[set] = dyS[(2 * j + half) * kLd + n32];
      fa1[set] = dyS[(2 * j + half) * kLd + 32 + n32];
      // two pixels on (clamped at the image end: those pixels' dy is zero)
      ow += 2;
      const bool cw = ow >= Wo;
      ow -= cw ? Wo : 0;
      oh += cw;
      const bool ce = oh * Wo + ow >= hw;
      oh = ce ? oh0 : oh;
      ow = ce ? 0 : ow;
    };
    frag(0, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int cur = j & 1;
      if (j + 1 < 16) frag(j + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const float bB = okB ? fbB[cur] : 0.f;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], fbA[cur], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cur], bB, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], fbA[cur], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cur], bB, acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the four waves' sums, in wave order, then the workgroup's row / atomics.  acc[mt][nt][e]: channel 32 mt + (e & 3) +
  // 8 (e >> 2) + 4 half, tap 32 nt + n32
  __syncthreads();
  float* red = smem7;  // [4][64][49] (50 KB: the patches are dead)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * half, tap = 32 * nt + n32;
        if (tap < kTaps) red[(wv * kS7C + c) * kTaps + tap] = acc[mt][nt][e];
      }
  __syncthreads();
  for (int i = threadIdx.x; i < kS7C * kTaps; i += kBlock) {
    float a = 0.f;
    for (int w_ = 0; w_ < kBlock / kWave; ++w_) a += red[w_ * kS7C * kTaps + i];
    if (partial) partial[(size_t)blockIdx.x * kS7C * kTaps + i] = a;
    else atomicAdd(dw + i, a);
  }
}

// dy = ga*(g-gmean) + gb*(y-mean): the gradient w.r.t. a conv output through its BatchNorm, written once for the
// convolution's weight and data gradients (both then read 4 instead of 8 bytes per element, the data gradient nine times).
// Written as the fp16-split GEMMs consume it (pwconv_f16.hip): two fp16 planes [rows][C] - h = fp16(dy S), then l = fp16(dy S - h),
// S = pow2_scale(bn[TTK_BN_AUX][TTK_AUX_DY_BOUND]) - the same 4 bytes per element as fp32, and the GEMM producers move them
// without arithmetic.  thread = (row, 8 channels): one 16-byte store per plane.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& l) {
  const f16x2_t hh = __builtin_convertvector(f32x2_t{a, b}, f16x2_t);
  const f32x2_t back = __builtin_convertvector(hh, f32x2_t);
  const f16x2_t ll = __builtin_convertvector(f32x2_t{a - back.x, b - back.y}, f16x2_t);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}
__global__ void __launch_bounds__(kBlock) bn_bwd_apply_k(const float* __restrict__ g, const float* __restrict__ y,
                                                          const float* __restrict__ bnp, uint16_t* __restrict__ dy, int64_t items, int C) {
  const int octs = C >> 3;
  const int c8 = threadIdx.x & (octs - 1);
  const float s = pow2_scale(bnp[(size_t)TTK_BN_AUX * C + TTK_AUX_DY_BOUND]);
  float ga[8], gb[8], gmean[8], mean[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ga[j] = bnp[TTK_BN_GA * C + 8 * c8 + j] * s;
    gb[j] = bnp[TTK_BN_GB * C + 8 * c8 + j] * s;
    gmean[j] = bnp[TTK_BN_GMEAN * C + 8 * c8 + j];
    mean[j] = bnp[TTK_BN_MEAN * C + 8 * c8 + j];
  }
  uint16_t* lo = dy + (size_t)items * 8;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kBlock) {
    const size_t off = (size_t)idx << 3;
    const float4 g0 = ld4(g + off), g1 = ld4(g + off + 4);  // g is read again (shortcut branch), y is not
    const float4 y0 = ld4nt(y + off), y1 = ld4nt(y + off + 4);
    const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ga[j] * (gv[j] - gmean[j]) + gb[j] * (yv[j] - mean[j]);
    uint4 h, l;
    split2(v[0], v[1], h.x, l.x); split2(v[2], v[3], h.y, l.y); split2(v[4], v[5], h.z, l.z); split2(v[6], v[7], h.w, l.w);
    *reinterpret_cast<uint4*>(dy + off) = h;
    *reinterpret_cast<uint4*>(lo + off) = l;
  }
}

// gs = (ga (+ gb)) * [a > 0]: gradient w.r.t. s = bn(y) + r of a block whose output activation is a = relu(s).
// part: BatchNorm-backward sums of bn(y) (sum gs, sum gs*(y-mean)); partd (with yd, bnd): the same for the
// downsample branch's BatchNorm.
// (256 threads: with 1024 - which pays in maxpool_bwd_k, one large tensor - the eight calls of a step lose 0.2 ms in total: the
// small late layers do not fill such workgroups and pay their 16-wave reduction tail twice)
constexpr int kEwBlock = kBlock;
__global__ void __launch_bounds__(kEwBlock) residual_bwd_k(const float* __restrict__ ga, const float* __restrict__ gb,
                                                          const float* __restrict__ a, const float* __restrict__ y,
                                                          float* __restrict__ bnp, const float* __restrict__ yd,
                                                          float* __restrict__ bnd, float* __restrict__ gs,
                                                          float* __restrict__ part, float* __restrict__ partd, int64_t items,
                                                          int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int quads = C >> 2;
  const int c4 = threadIdx.x & (quads - 1);
  const float4 mean = ld4(bnp + TTK_BN_MEAN * C + 4 * c4);
  const float4 meand = yd ? ld4(bnd + TTK_BN_MEAN * C + 4 * c4) : f4(0.f);
  float4 s1 = f4(0.f), s2 = f4(0.f), t2 = f4(0.f);
  float gmx = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * kEwBlock + threadIdx.x; idx < items; idx += (int64_t)gridDim.x * kEwBlock) {
    const size_t off = (size_t)idx << 2;
    float4 gv = ld4(ga + off);
    if (gb) gv = add4(gv, ld4(gb + off));
    gv = mask4(gv, ld4(a + off));
    st4(gs + off, gv);
    gmx = fmaxf(gmx, fmaxf(fmaxf(fabsf(gv.x), fabsf(gv.y)), fmaxf(fabsf(gv.z), fabsf(gv.w))));
    s1 = add4(s1, gv);
    s2 = fma4(gv, sub4(ld4(y + off), mean), s2);
    if (yd) t2 = fma4(gv, sub4(ld4(yd + off), meand), t2);
  }
  // max |gs|: the bound behind the DY_BOUND of both BatchNorms this gradient flows into (ttk.h, TTK_AUX_GMAX)
  wave_raise_max(bnp + (size_t)TTK_BN_AUX * C + TTK_AUX_GMAX, gmx);
  if (yd) wave_raise_max(bnd + (size_t)TTK_BN_AUX * C + TTK_AUX_GMAX, gmx);
  block_channel_partials<1024, kEwBlock>(s1, s2, c4, C, part + (size_t)blockIdx.x * 2 * C, smem);
  if (yd) {
    __syncthreads();
    block_channel_partials<1024, kEwBlock>(s1, t2, c4, C, partd + (size_t)blockIdx.x * 2 * C, smem);
  }
}

static bool ew_shape_ok(int64_t rows, int C) { return rows > 0 && C >= 32 && C <= 1024 && (C & (C - 1)) == 0; }


// ---- BlurPool2D on channels-last rows (reference neuralnets/modelcomponents.py:187-205: the ResNet variant's use_blurpool,
// backbones/resnet.py:31-49,63-66): depthwise 3x3 with the binomial kernel [1 2 1]^T [1 2 1] / 16, zero padding 1, stride 1 | 2.
// One thread = one pixel x 4 channels; HBM-bound and small beside the dense convolutions around it.
__device__ __forceinline__ float blur_tap(int k) { return k == 1 ? 0.5f : 0.25f; }  // separable: (1/4, 1/2, 1/4) per axis

__global__ void __launch_bounds__(kBlock) blur3x3_fwd_k(const float* __restrict__ a, float* __restrict__ t, int B, int H, int W, int Ho, int Wo,
                                                        int C, int stride) {
  const int cq = C / 4;
  const int64_t items = (int64_t)B * Ho * Wo * cq;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < items; i += (int64_t)gridDim.x * kBlock) {
    const int q = (int)(i % cq);
    int64_t px = i / cq;
    const int wo = (int)(px % Wo);
    px /= Wo;
    const int ho = (int)(px % Ho), n = (int)(px / Ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = ho * stride + kh - 1;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = wo * stride + kw - 1;
        if (wi < 0 || wi >= W) continue;
        const float wgt = blur_tap(kh) * blur_tap(kw);
        const float4 v = ld4(a + (((int64_t)n * H + hi) * W + wi) * C + 4 * q);
        acc.x = fmaf(wgt, v.x, acc.x); acc.y = fmaf(wgt, v.y, acc.y); acc.z = fmaf(wgt, v.z, acc.z); acc.w = fmaf(wgt, v.w, acc.w);
      }
    }
    st4(t + i * 4, acc);
  }
}

// g[n][hi][wi] = sum over (ho, wo, kh, kw) with ho*stride + kh - 1 = hi, wo*stride + kw - 1 = wi of w[kh][kw] * (ga + gb)[n][ho][wo]
__global__ void __launch_bounds__(kBlock) blur3x3_bwd_k(const float* __restrict__ ga, const float* __restrict__ gb, float* __restrict__ g, int B,
                                                        int H, int W, int Ho, int Wo, int C, int stride) {
  const int cq = C / 4;
  const int64_t items = (int64_t)B * H * W * cq;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < items; i += (int64_t)gridDim.x * kBlock) {
    const int q = (int)(i % cq);
    int64_t px = i / cq;
    const int wi = (int)(px % W);
    px /= W;
    const int hi = (int)(px % H), n = (int)(px / H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int th = hi + 1 - kh;
      if (th < 0 || th % stride != 0) continue;
      const int ho = th / stride;
      if (ho >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tw = wi + 1 - kw;
        if (tw < 0 || tw % stride != 0) continue;
        const int wo = tw / stride;
        if (wo >= Wo) continue;
        const float wgt = blur_tap(kh) * blur_tap(kw);
        const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * C + 4 * q;
        float4 v = ld4(ga + o);
        if (gb) {
          const float4 u = ld4(gb + o);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        acc.x = fmaf(wgt, v.x, acc.x); acc.y = fmaf(wgt, v.y, acc.y); acc.z = fmaf(wgt, v.z, acc.z); acc.w = fmaf(wgt, v.w, acc.w);
      }
    }
    st4(g + i * 4, acc);
  }
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_stem7_fwd(const float* x, const float* w, float* y, float* part, const float* pivot, int B, int H, int W, ttk_stream_t stream) {
  TTK_REQUIRE(x && w && y, "stem7_fwd: null pointer");
  TTK_REQUIRE(B > 0 && H > 6 && W > 6, "stem7_fwd: bad shape B=%d H=%d W=%d", B, H, W);
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int64_t items = (int64_t)B * Ho * Wo * (kS7C / 4);  // sizes the partial rows (ttk_partial_rows_elementwise)
  const int PR = 2 * ((Wo + 30) / Wo) + 7;          // input rows under 32 consecutive output pixels
  const int prows = (PR + 8) / 9 * 9;               // staged in rounds of 9
  const int wave_floats = prows * kS7Wp > 32 * (kS7C + 4) ? prows * kS7Wp : 32 * (kS7C + 4);
  const size_t smem = (size_t)(kBlock / kWave) * wave_floats * sizeof(float);
  static const bool valu = exp_env("TTK_STEM7_VALU") != nullptr;  // the previous kernel (A/B timing)
  if (!valu && smem <= 60 * 1024 && W + 7 <= kS7Wp && (int64_t)B * ((Ho * Wo + 31) / 32) < ((int64_t)1 << 30))
    hipLaunchKernelGGL(stem7_fwd_mfma_k, dim3(elementwise_grid(items)), dim3(kBlock), smem, (hipStream_t)stream, x, w, y, part, pivot, B, H, W,
                       Ho, Wo, PR, wave_floats);
  else
    hipLaunchKernelGGL(stem7_fwd_k, dim3(elementwise_grid(items), kS7C / kS7Half), dim3(kBlock), 0, (hipStream_t)stream, x, w, y, part,
                       pivot, B, H, W, Ho, Wo);
  TTK_LAUNCH_CHECK("stem7_fwd");
}

size_t ttk_stem7_wgrad_partial_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 6 || W <= 6) return 0;
  return (size_t)stem_wgrad_grid(B, (H + 6 - 7) / 2 + 1) * kS7C * kS7K * kS7K * sizeof(float);
}

int ttk_stem7_bwd_weight(const float* g, const float* y, const float* bn, const float* x, float* dw, float* partial, int B, int H, int W,
                         ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn && x && dw, "stem7_bwd_weight: null pointer");
  TTK_REQUIRE(B > 0 && H > 6 && W > 6, "stem7_bwd_weight: bad shape");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  (void)hipMemsetAsync(dw, 0, sizeof(float) * kS7C * kS7K * kS7K, (hipStream_t)stream);
  const int PR = 2 * ((Wo + 30) / Wo) + 7, prows = (PR + 8) / 9 * 9;
  const int patch_floats = prows * kS7WpW, wave_floats = patch_floats + 32 * (kS7C + 4);
  size_t smem = (size_t)(kBlock / kWave) * wave_floats * sizeof(float);
  const size_t red_bytes = (size_t)(kBlock / kWave) * kS7C * kS7K * kS7K * sizeof(float);
  if (smem < red_bytes) smem = red_bytes;
  static const bool valu = exp_env("TTK_STEM7_VALU") != nullptr;  // the previous kernels (A/B timing)
  if (!valu && smem <= 64 * 1024 && W + 7 <= 192 && (int64_t)B * ((Ho * Wo + 31) / 32) < ((int64_t)1 << 30)) {
    const int grid = stem_wgrad_grid(B, Ho);
    hipLaunchKernelGGL(stem7_wgrad_mfma_k, dim3(grid), dim3(kBlock), smem, (hipStream_t)stream, g, y, bn, x, dw, partial, B, H, W, Ho, Wo, PR,
                       patch_floats, wave_floats);
    if (partial) launch_fold_partials(partial, grid, (int64_t)kS7C * kS7K * kS7K, dw, 1, (hipStream_t)stream);
  } else {
    launch_stem_wgrad<kS7K, kS7C>(g, y, bn, x, dw, partial, B, H, W, Ho, Wo, (hipStream_t)stream);
  }
  TTK_LAUNCH_CHECK("stem7_bwd_weight");
}

int ttk_maxpool3x3s2_fwd(const float* y, float* bn, float* a, unsigned char* idx, int B, int H, int W, int C,
                         ttk_stream_t stream) {
  TTK_REQUIRE(y && bn && a && idx, "maxpool3x3s2_fwd: null pointer");
  TTK_REQUIRE(B > 0 && H > 1 && W > 1 && ew_shape_ok(1, C), "maxpool3x3s2_fwd: unsupported shape");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t items = (int64_t)B * Ho * Wo * (C / 4);
  TTK_REQUIRE((int64_t)B * H * W * (C / 4) < (int64_t)1 << 31, "maxpool3x3s2_fwd: tensor too large for 32-bit indexing");
  int64_t grid = ceil_div(items, kBlock);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(maxpool_fwd_k, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, y, bn, a, idx, B, H, W, Ho, Wo, C);
  TTK_LAUNCH_CHECK("maxpool3x3s2_fwd");
}

int ttk_maxpool3x3s2_bwd(const float* ga, const float* gb, const unsigned char* idx, const float* y, const float* bn, float* g,
                         float* part, int B, int H, int W, int C, ttk_stream_t stream) {
  TTK_REQUIRE(ga && idx && y && bn && g, "maxpool3x3s2_bwd: null pointer");
  TTK_REQUIRE(B > 0 && H > 1 && W > 1 && ew_shape_ok(1, C), "maxpool3x3s2_bwd: unsupported shape");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t items = (int64_t)B * H * W * (C / 4);
  TTK_REQUIRE(items < (int64_t)1 << 31, "maxpool3x3s2_bwd: tensor too large for 32-bit indexing");
  hipLaunchKernelGGL(maxpool_bwd_k, dim3(elementwise_grid(items)), dim3(1024), 2 * (size_t)C * sizeof(float), (hipStream_t)stream,
                     ga, gb, idx, y, bn, g, part, B, H, W, Ho, Wo, C);
  TTK_LAUNCH_CHECK("maxpool3x3s2_bwd");
}

int ttk_blur3x3_fwd(const float* a, float* t, int B, int H, int W, int C, int stride, ttk_stream_t stream) {
  TTK_REQUIRE(a && t, "blur3x3_fwd: null pointer");
  TTK_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "blur3x3_fwd: unsupported shape");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  int64_t grid = ceil_div((int64_t)B * Ho * Wo * (C / 4), kBlock);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(blur3x3_fwd_k, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, a, t, B, H, W, Ho, Wo, C, stride);
  TTK_LAUNCH_CHECK("blur3x3_fwd");
}

int ttk_blur3x3_bwd(const float* ga, const float* gb, float* g, int B, int H, int W, int C, int stride, ttk_stream_t stream) {
  TTK_REQUIRE(ga && g, "blur3x3_bwd: null pointer");
  TTK_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "blur3x3_bwd: unsupported shape");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  int64_t grid = ceil_div((int64_t)B * H * W * (C / 4), kBlock);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(blur3x3_bwd_k, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, ga, gb, g, B, H, W, Ho, Wo, C, stride);
  TTK_LAUNCH_CHECK("blur3x3_bwd");
}

int ttk_bn_add_act(const float* y, float* bn, const float* res, const float* res_bn, float* a, const float* res_bound, int measure,
                   int64_t rows, int C, ttk_stream_t stream) {
  TTK_REQUIRE(y && bn && a && (res || !res_bn), "bn_add_act: bad arguments");
  TTK_REQUIRE(ew_shape_ok(rows, C), "bn_add_act: unsupported shape rows=%lld C=%d", (long long)rows, C);
  const int64_t items = rows * (C / 4);
  int64_t grid = ceil_div(items, kBlock);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(bn_add_act_k, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, y, bn, res, res_bn, a, res_bound, measure, items, C);
  TTK_LAUNCH_CHECK("bn_add_act");
}

int ttk_bn_bwd_apply(const float* g, const float* y, const float* bn, void* dy, int64_t rows, int C, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn && dy, "bn_bwd_apply: null pointer");
  TTK_REQUIRE(ew_shape_ok(rows, C), "bn_bwd_apply: unsupported shape rows=%lld C=%d", (long long)rows, C);
  const int64_t items = rows * (C / 8);
  int64_t grid = ceil_div(items, kBlock);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(bn_bwd_apply_k, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, g, y, bn, (uint16_t*)dy, items, C);
  TTK_LAUNCH_CHECK("bn_bwd_apply");
}

int ttk_residual_bwd(const float* ga, const float* gb, const float* a, const float* y, float* bn, const float* yd,
                     float* bnd, float* gs, float* part, float* partd, int64_t rows, int C, ttk_stream_t stream) {
  TTK_REQUIRE(ga && a && y && bn && gs && part, "residual_bwd: null pointer");
  TTK_REQUIRE((yd == nullptr) == (bnd == nullptr) && (yd == nullptr) == (partd == nullptr), "residual_bwd: yd, bnd, partd go together");
  TTK_REQUIRE(ew_shape_ok(rows, C), "residual_bwd: unsupported shape");
  const int64_t items = rows * (C / 4);
  hipLaunchKernelGGL(residual_bwd_k, dim3(elementwise_grid(items)), dim3(kEwBlock), 2 * (size_t)C * sizeof(float), (hipStream_t)stream,
                     ga, gb, a, y, bn, yd, bnd, gs, part, partd, items, C);
  TTK_LAUNCH_CHECK("residual_bwd");
}

}  // extern "C"
