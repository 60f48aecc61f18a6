// Stem convolution 5x5, stride 2, pad 2, 1 -> 32 channels (backbones/mobilenet_v1.py:122-124,161)
// and its weight gradient.  HBM-bound: the output (B*65*65*32 floats) dominates; the 129x129 input
// plane is read through L1/L2 (each pixel is touched by <= 9 outputs x 8 lanes).
#include <stdint.h>

#include "ttk_common.h"
#include <cstdlib>
#include <cstring>

namespace ttk {

constexpr int kStemC = 32;
constexpr int kStemQuads = kStemC / 4;

#if defined(TTK_EXPERIMENTS)  // round 1's VALU kernels: selectable by TTK_STEM=scalar only (A/B timing)
// thread = TWO horizontally adjacent output pixels, all 32 channels.  The filter bank sits in LDS as
// wt[tap][c]; a ds_read_b128 at a wave-uniform address is a broadcast and hands every lane the weights of 4
// channels for one tap, which feed 8 FMAs (2 pixels x 4 channels).  35 coalesced input loads (5 rows x 7
// columns, shared by the two pixels) -> 1600 FMAs -> 2 x 128 B of output per thread.
// v1 (thread = pixel x channel quad: one LDS read per 4 FMAs, inputs re-loaded by 8 lanes) ran at 0.76 TB/s;
// a scalar-register filter bank does not work either (800 SGPRs: the compiler spills them through
// v_writelane/v_readlane).  BatchNorm partial sums stay in registers over the grid-stride loop.
__global__ void __launch_bounds__(kBlock) stem_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ y, float* __restrict__ part, const float* __restrict__ pivot,
                                                      int B, int H, int W, int Ho, int Wo) {
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
      const float4 pv = pivot ? ld4(pivot + 4 * c4) : f4(0.f);  // the sums are those of y - pivot
      a0 = sub4(a0, pv);
      if (second) { st4(y0 + kStemC + 4 * c4, a1); a1 = sub4(a1, pv); }
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
#endif  // TTK_EXPERIMENTS


// ---------------------------------------------------------------------------------------------------------------
// The same two operations on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: true fp32 products and accumulation, the
// instruction the early pointwise layers use).  The scalar kernels above spend 1600 FMAs + 200 LDS broadcasts per
// thread-iteration and run at 1.6 TB/s (forward) / 2.3 TB/s (weight gradient); as GEMMs the arithmetic is ~25 us and
// the kernels become what the header says they are, HBM streams.
//
// forward:  y[px][c] = sum_t patch[px][t] * w[c][t]            M = 32 pixels, N = 32 channels, K = 25 taps (13 k-pairs)
// wgrad:    dW[c][t] = sum_px dy[px][c] * patch[px][t]         M = 32 channels, N = 32 (25 taps + 7 idle), K = pixels
// A fragment: lane l supplies A[row = l % 32][k = l / 32]; B fragment: B[k = l / 32][col = l % 32]; the accumulator of
// lane l holds column l % 32, rows (e & 3) + 8 * (e >> 2) + 4 * (l / 32), e = 0..15.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float stem_patch(const float* __restrict__ x, int n, int ho, int wo, int tap, int H, int W, bool ok) {
  const int kh = tap / 5, kw = tap - 5 * kh;
  const int hi = 2 * ho + kh - 2, wi = 2 * wo + kw - 2;
  return (ok && tap < 25 && hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((size_t)n * H + hi) * W + wi] : 0.f;
}

template <typename T>
__global__ void __launch_bounds__(kBlock) stem_fwd_mfma_k(const float* __restrict__ x, const float* __restrict__ w,
                                                           T* __restrict__ y, float* __restrict__ part, const float* __restrict__ pivot,
                                                           int B, int H, int W, int Ho, int Wo) {
  __shared__ float red[kBlock / kWave][2 * kStemC];
  constexpr int kTileLd = 36;  // floats per pixel row of the transposition tile (16-byte aligned rows, conflict-free b128 reads)
  __shared__ __attribute__((aligned(16))) float tile[kBlock / kWave][32 * kTileLd];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, hk = lane >> 5;
  const float pv = pivot ? pivot[r] : 0.f;  // the sums are those of y - pivot
  float bw[13];  // B fragments: w[c = r][tap = 2j + hk]
#pragma unroll
  for (int j = 0; j < 13; ++j) bw[j] = (2 * j + hk < 25) ? w[r * 25 + 2 * j + hk] : 0.f;
  float s1 = 0.f, s2 = 0.f;  // column r, rows of this lane's half
  const int64_t P = (int64_t)B * Ho * Wo, tiles = (P + 31) / 32;
  const int HoWo = Ho * Wo;
  // the patch of the NEXT tile is requested before this tile's MFMAs and stores (a wave had nothing in flight while it
  // multiplied and wrote; four waves per SIMD do not cover a dependent chain of gather -> 13 MFMAs -> transpose -> store).
  // 105 -> 99 us at B = 512 with the 16-byte stores below.  Timing-only builds: without the patch loads 62 us, without the
  // stores 82, without the MFMAs 90 - the 13 strided dword gathers per lane are what is left to remove (an LDS-staged input
  // band as in the weight gradient).
  const int64_t tstep = (int64_t)gridDim.x * (kBlock / kWave);
  float an[13];
  auto load_patch = [&](int64_t t) {
    const int64_t px = t * 32 + r;
    const bool ok = px < P;
    const unsigned pxu = ok ? (unsigned)px : 0u;  // (B * Ho * Wo < 2^31: checked by the host)
    const int n = (int)(pxu / (unsigned)HoWo), rem = (int)(pxu - (unsigned)n * (unsigned)HoWo), ho = rem / Wo, wo = rem - ho * Wo;
#pragma unroll
    for (int j = 0; j < 13; ++j) an[j] = stem_patch(x, n, ho, wo, 2 * j + hk, H, W, ok);
  };
  const int64_t t0 = (int64_t)blockIdx.x * (kBlock / kWave) + wv;
  if (t0 < tiles) load_patch(t0);
  for (int64_t t = t0; t < tiles; t += tstep) {
    float a[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) a[j] = an[j];
    if (t + tstep < tiles) load_patch(t + tstep);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 13; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bw[j], acc, 0, 0, 0);
    if constexpr (!Act<T>::kBf16) {
      // fp32 storage: the tile goes through a wave-private LDS transpose and leaves as four 16-byte stores per lane (1 KB
      // contiguous per instruction) instead of sixteen 4-byte stores
      float* tw = &tile[wv][0];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * hk;
        tw[row * kTileLd + r] = acc[e];
        if (t * 32 + row < P) {
          const float d = acc[e] - pv;
          s1 += d;
          s2 = fmaf(d, d, s2);
        }
      }
      __builtin_amdgcn_wave_barrier();  // wave-private: LDS executes a wave's accesses in order
      float* yt4 = reinterpret_cast<float*>(y) + (size_t)t * 32 * kStemC;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i, px_ = q >> 3, c4 = q & 7;
        const float4 v = ld4(tw + px_ * kTileLd + 4 * c4);
        if (t * 32 + px_ < P) st4(yt4 + (size_t)q * 4, v);
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      T* yt = y + (size_t)t * 32 * kStemC + r;  // column r of the tile's 32 pixels
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * hk;
        if (t * 32 + row < P) {
          const float v = Act<T>::st1(yt + (size_t)row * kStemC, acc[e]) - pv;  // 32 lanes = one pixel; statistics of the stored value
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      }
    }
  }
  if (part) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lane < 32) { red[wv][r] = s1; red[wv][kStemC + r] = s2; }
    __syncthreads();
    if (threadIdx.x < 2 * kStemC) {
      float a = 0.f;
      for (int i = 0; i < kBlock / kWave; ++i) a += red[i][threadIdx.x];  // fixed wave order
      part[(size_t)blockIdx.x * 2 * kStemC + threadIdx.x] = a;
    }
  }
}

// Forward with the input band in LDS (default): a workgroup owns a band of kFwBand output rows of one image, stages the
// 2 kFwBand + 3 input rows it touches once with coalesced loads (two zero columns left and right, zero rows outside the image:
// no bounds tests at the taps), and its waves take the band's pixels 32 at a time - the patch fragments are LDS reads instead
// of 13 strided 4-byte global gathers per lane (timing-only builds of stem_fwd_mfma_k: 99 us with them, 62 without).
constexpr int kFwBand = 13;  // LDS is sized for bands of up to 13 output rows (65 = 5 x 13); the launch picks the height, see stem_band_rows()
template <typename T>
__global__ void __launch_bounds__(kBlock) stem_fwd_band_k(const float* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                           float* __restrict__ part, const float* __restrict__ pivot, int B, int H, int W, int Ho,
                                                           int Wo, int nbands, int band_rows) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[kBlock / kWave][2 * kStemC];
  constexpr int kTileLd = 36;
  const int Wp = W + 4, xrows = 2 * kFwBand + 3;  // (layout of the LDS carve-up: the staged rows of a band are 2 band_rows + 3 <= xrows)
  float* xs = sm;                                  // [xrows][Wp]
  float* tiles = sm + ((xrows * Wp + 3) & ~3);     // [4 waves][32][kTileLd]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, hk = lane >> 5;
  const float pv = pivot ? pivot[r] : 0.f;  // the sums are those of y - pivot
  float bw[13];
  int toff[13];  // LDS offset of this lane's tap 2j + hk inside a patch
#pragma unroll
  for (int j = 0; j < 13; ++j) {
    const int tap = 2 * j + hk, kh = tap / 5, kw = tap - 5 * kh;
    bw[j] = tap < 25 ? w[r * 25 + tap] : 0.f;
    toff[j] = tap < 25 ? kh * Wp + kw : 0;
  }
  float* tw = tiles + wv * 32 * kTileLd;
  float s1 = 0.f, s2 = 0.f;
  for (int bt = blockIdx.x; bt < B * nbands; bt += gridDim.x) {
    const int n = bt / nbands, band = bt - n * nbands;
    const int ho0 = band * band_rows, ho1 = min(ho0 + band_rows, Ho), hi0 = 2 * ho0 - 2;
    const int nrows = 2 * (ho1 - ho0) + 3;
    __syncthreads();  // the previous band's readers are done with xs
    // eight loads per thread in flight: one by one this loop was a chain of ~15 memory round trips between two barriers
    for (int i0 = threadIdx.x; i0 < nrows * Wp; i0 += 8 * kBlock) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * kBlock, rr = i / Wp, cc = i - rr * Wp, hi = hi0 + rr, wi = cc - 2;
        v[u] = (i < nrows * Wp && hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((size_t)n * H + hi) * W + wi] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * kBlock < nrows * Wp) xs[i0 + u * kBlock] = v[u];
    }
    __syncthreads();
    const int npix = (ho1 - ho0) * Wo;
    T* yb = y + ((size_t)n * Ho + ho0) * Wo * kStemC;  // the band's pixels are contiguous in y
    for (int p0 = wv * 32; p0 < npix; p0 += (kBlock / kWave) * 32) {
      const int p = p0 + r < npix ? p0 + r : 0;  // lanes past the band compute pixel 0 again and store nothing
      const int hol = p / Wo, wo = p - hol * Wo;
      const float* base = xs + (2 * hol) * Wp + 2 * wo;
      float a[13];
#pragma unroll
      for (int j = 0; j < 13; ++j) a[j] = base[toff[j]];
      if (hk) a[12] = 0.f;  // tap 25 does not exist
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 13; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bw[j], acc, 0, 0, 0);
      if constexpr (!Act<T>::kBf16) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * hk;
          tw[row * kTileLd + r] = acc[e];
          if (p0 + row < npix) {
            const float d = acc[e] - pv;
            s1 += d;
            s2 = fmaf(d, d, s2);
          }
        }
        __builtin_amdgcn_wave_barrier();  // wave-private tile: LDS executes a wave's accesses in order
        float* yt4 = reinterpret_cast<float*>(yb) + (size_t)p0 * kStemC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = lane + 64 * i, px_ = q >> 3, c4 = q & 7;
          const float4 v = ld4(tw + px_ * kTileLd + 4 * c4);
          if (p0 + px_ < npix) st4(yt4 + (size_t)q * 4, v);
        }
        __builtin_amdgcn_wave_barrier();
      } else {
        T* yt = yb + (size_t)p0 * kStemC + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * hk;
          if (p0 + row < npix) {
            const float v = Act<T>::st1(yt + (size_t)row * kStemC, acc[e]) - pv;  // statistics of the stored value
            s1 += v;
            s2 = fmaf(v, v, s2);
          }
        }
      }
    }
  }
  if (part) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lane < 32) { red[wv][r] = s1; red[wv][kStemC + r] = s2; }
    __syncthreads();
    if (threadIdx.x < 2 * kStemC) {
      float a = 0.f;
      for (int i = 0; i < kBlock / kWave; ++i) a += red[i][threadIdx.x];  // fixed wave order
      part[(size_t)blockIdx.x * 2 * kStemC + threadIdx.x] = a;
    }
  }
}

// Weight gradient.  A workgroup owns a band of kWgBand output rows of one image: the input rows the band touches sit in
// LDS (zero outside the image), so the patch fragments are LDS reads; dy is loaded with 16-byte loads (512 B per
// wave-load instead of the fragment layout's 4 bytes per lane - the dword version was texture-addresser bound at
// 255 us), formed in registers and turned into fragment order through a wave-private LDS tile.
constexpr int kWgBand = 13;   // LDS is sized for bands of up to 13 output rows; the launch picks the height (stem_band_rows)
constexpr int kWgChunk = 32;  // pixels per wave iteration = 16 MFMA k-pairs (four 16-byte loads per lane and tensor in flight)

template <typename T, typename TG>
__global__ void __launch_bounds__(kBlock) stem_wgrad_mfma_k(const TG* __restrict__ g, const T* __restrict__ y,
                                                             const float* __restrict__ bn, const float* __restrict__ x,
                                                             float* __restrict__ dw, float* __restrict__ partial, int B, int H, int W, int Ho, int Wo,
                                                             int nbands, int band_rows) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int xrows = 2 * kWgBand + 3;
  float* xs = sm;                                         // [xrows][W]
  float* dys = sm + xrows * W;                            // [4 waves][kWgChunk][32]
  float* red = dys + (kBlock / kWave) * kWgChunk * kStemC;  // [4 waves][32][25]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, hk = lane >> 5;
  const int c4 = lane & 7, pl = lane >> 3;  // staging role: channel quad, pixel 0..7 (+8)
  const float4 ga = ld4(bn + TTK_BN_GA * kStemC + 4 * c4), gb = ld4(bn + TTK_BN_GB * kStemC + 4 * c4),
               gmean = ld4(bn + TTK_BN_GMEAN * kStemC + 4 * c4), mean = ld4(bn + TTK_BN_MEAN * kStemC + 4 * c4);
  const int kh = r / 5, kw = r - 5 * kh;  // this lane's tap (r < 25)
  float* dyw = dys + wv * kWgChunk * kStemC;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int t = blockIdx.x; t < B * nbands; t += gridDim.x) {
    const int n = t / nbands, band = t - n * nbands;
    const int ho0 = band * band_rows, ho1 = min(ho0 + band_rows, Ho);
    const int hi0 = 2 * ho0 - 2;
    __syncthreads();  // previous band's readers are done with xs
    // eight loads per thread in flight (see stem_fwd_band_k)
    for (int i0 = threadIdx.x; i0 < xrows * W; i0 += 8 * kBlock) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * kBlock, rr = i / W, cc = i - rr * W, hi = hi0 + rr;
        v[u] = (i < xrows * W && hi >= 0 && hi < H) ? x[((size_t)n * H + hi) * W + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * kBlock < xrows * W) xs[i0 + u * kBlock] = v[u];
    }
    __syncthreads();
    const int npix = (ho1 - ho0) * Wo;
    const size_t pix0 = ((size_t)n * Ho + ho0) * Wo;  // the band's pixels are contiguous in g / y
    // the loads of the NEXT chunk are issued before this chunk's MFMAs (the wave had nothing in flight while it multiplied:
    // 153 -> 146 us at B = 512)
    float4 gvs[kWgChunk / 8], yvs[kWgChunk / 8];
    auto load_chunk = [&](int p0) {
#pragma unroll
      for (int u = 0; u < kWgChunk / 8; ++u) {
        const int pp = p0 + pl + 8 * u;
        const size_t o = (pix0 + (pp < npix ? pp : 0)) * kStemC + 4 * c4;
        gvs[u] = Act<TG>::ldnt(g + o);
        yvs[u] = Act<T>::ldnt(y + o);
      }
    };
    if (wv * kWgChunk < npix) load_chunk(wv * kWgChunk);
    for (int p0 = wv * kWgChunk; p0 < npix; p0 += (kBlock / kWave) * kWgChunk) {
      // ---- dy of 32 pixels x 32 channels: four 16-byte loads per lane and tensor (requested one chunk earlier)
#pragma unroll
      for (int u = 0; u < kWgChunk / 8; ++u) {
        const int pp = p0 + pl + 8 * u;
        float4 d = f4(0.f);
        if (pp < npix) {
          const float4 gv = gvs[u], yv = yvs[u];
          d = make_float4(fmaf(ga.x, gv.x - gmean.x, gb.x * (yv.x - mean.x)), fmaf(ga.y, gv.y - gmean.y, gb.y * (yv.y - mean.y)),
                          fmaf(ga.z, gv.z - gmean.z, gb.z * (yv.z - mean.z)), fmaf(ga.w, gv.w - gmean.w, gb.w * (yv.w - mean.w)));
        }
        st4(dyw + (pl + 8 * u) * kStemC + 4 * c4, d);
      }
      if (p0 + (kBlock / kWave) * kWgChunk < npix) load_chunk(p0 + (kBlock / kWave) * kWgChunk);
      __builtin_amdgcn_wave_barrier();  // the tile is wave-private; LDS executes a wave's accesses in order
      // pixel of this lane's k index: p0 + 2j + hk; its (row, column) inside the band advance without divisions
      int ho = (p0 + hk) / Wo, wo = (p0 + hk) - ho * Wo;
      const float* xrow = xs + (2 * ho + kh) * W + kw - 2;
#pragma unroll
      for (int j = 0; j < kWgChunk / 2; ++j) {
        const int q = 2 * j + hk;
        const float a = dyw[q * kStemC + r];
        const int wi = 2 * wo + kw - 2;
        const float b = (r < 25 && p0 + q < npix && wi >= 0 && wi < W) ? xrow[2 * wo] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        wo += 2;
        if (wo >= Wo) { wo -= Wo; xrow += 2 * W; }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // lane holds dW[c = row(e, hk)][tap = r]; fold the waves of the block, then one atomic per weight and block
  if (r < 25) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wv * kStemC + (e & 3) + 8 * (e >> 2) + 4 * hk) * 25 + r] = acc[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kStemC * 25; i += kBlock) {
    float v = 0.f;
    for (int q = 0; q < kBlock / kWave; ++q) v += red[q * kStemC * 25 + i];
    if (partial) partial[(size_t)blockIdx.x * (kStemC * 25) + i] = v;  // deterministic mode: folded by fold_partials_k
    else atomicAdd(dw + i, v);  // dw[c][tap]
  }
}

__global__ void zero_k(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

}  // namespace ttk

// Band height (output rows per tile) for `grid` persistent workgroups: the workgroup with the most tiles sets the kernel's time, so pick
// the height that minimises ceil(tiles / grid) * rows - at B = 512, Ho = 65: 11 rows = 6 bands = 3 072 tiles = 3 per workgroup of 1 024
// (13 rows: 2 560 tiles, 3 for half of the workgroups and 2 for the rest: 39 rows against 33).
static int stem_band_rows(int B, int Ho, int grid, int max_rows) {
  int best = max_rows < Ho ? max_rows : Ho;
  int64_t best_cost = INT64_MAX;
  for (int r = (max_rows < Ho ? max_rows : Ho); r >= 6 && r >= 1; --r) {
    const int64_t tiles = (int64_t)B * ((Ho + r - 1) / r);
    const int64_t cost = ttk::ceil_div(tiles, grid) * r;
    if (cost < best_cost) { best_cost = cost; best = r; }
  }
  return best;
}

using namespace ttk;

extern "C" {

int ttk_stem_fwd(const float* x, const float* w, void* y, float* part, const float* pivot, int B, int H, int W, int act_bf16,
                 ttk_stream_t stream) {
  TTK_REQUIRE(x && w && y, "stem_fwd: null pointer");
  TTK_REQUIRE(B > 0 && H > 4 && W > 4, "stem_fwd: bad shape B=%d H=%d W=%d", B, H, W);
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t items = (int64_t)B * Ho * Wo * kStemQuads;  // sizes the partial rows (ttk_partial_rows_elementwise)
  static const bool scalar = [] { const char* e = exp_env("TTK_STEM"); return e && strcmp(e, "scalar") == 0; }();
  TTK_REQUIRE(!(scalar && act_bf16), "stem_fwd: TTK_STEM=scalar has no bf16-storage form");
  TTK_REQUIRE((int64_t)B * Ho * Wo < (int64_t)1 << 31, "stem_fwd: too many output pixels for 32-bit indexing");
#if defined(TTK_EXPERIMENTS)
  if (scalar)  // TTK_STEM=scalar: the VALU kernels (A/B timing)
    hipLaunchKernelGGL(stem_fwd_k, dim3(elementwise_grid(items)), dim3(kBlock), 0, (hipStream_t)stream, x, w, (float*)y, part, pivot, B, H, W, Ho,
                       Wo);
  else
#endif
  {
    // input band in LDS (TTK_STEM=gather: the kernel that gathers its patches from global memory; also for images too wide for LDS)
    static const bool gather = [] { const char* e = exp_env("TTK_STEM"); return e && strcmp(e, "gather") == 0; }();
    const size_t sm = ((size_t)(((2 * kFwBand + 3) * (W + 4) + 3) & ~3) + (kBlock / kWave) * 32 * 36) * sizeof(float);
    const int band_rows = kFwBand < Ho ? kFwBand : Ho;  // (balanced 11-row bands measured no better here: 74.9 vs 71.1 us - more halo rows per output row)
    const int nbands = (Ho + band_rows - 1) / band_rows;
    if (!gather && sm <= 64 * 1024)
      TTK_ACT_DISPATCH_STEM(act_bf16, hipLaunchKernelGGL((stem_fwd_band_k<ActT>), dim3(elementwise_grid(items)), dim3(kBlock), sm, (hipStream_t)stream, x, w,
                                                    (ActT*)y, part, pivot, B, H, W, Ho, Wo, nbands, band_rows));
    else
      TTK_ACT_DISPATCH_STEM(act_bf16, hipLaunchKernelGGL((stem_fwd_mfma_k<ActT>), dim3(elementwise_grid(items)), dim3(kBlock), 0, (hipStream_t)stream, x, w,
                                                    (ActT*)y, part, pivot, B, H, W, Ho, Wo));
  }
  TTK_LAUNCH_CHECK("stem_fwd");
}

size_t ttk_stem_wgrad_partial_bytes(void) { return (size_t)1024 * 25 * kStemC * sizeof(float); }

int ttk_stem_bwd_weight(const void* g, const void* y, const float* bn, const float* x, float* dw, int accumulate, float* partial,
                        int B, int H, int W, int act_bf16, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn && x && dw, "stem_bwd_weight: null pointer");
  TTK_REQUIRE(B > 0 && H > 4 && W > 4, "stem_bwd_weight: bad shape");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t items = (int64_t)B * Ho * Wo * kStemQuads;
  int grid = elementwise_grid(items);
  static const bool scalar = [] { const char* e = exp_env("TTK_STEM"); return e && strcmp(e, "scalar") == 0; }();
  TTK_REQUIRE(!(scalar && act_bf16), "stem_bwd_weight: TTK_STEM=scalar has no bf16-storage form");
  if (scalar) partial = nullptr;
  if (!accumulate && !partial) hipLaunchKernelGGL(zero_k, dim3(4), dim3(256), 0, (hipStream_t)stream, dw, 25 * kStemC);
#if defined(TTK_EXPERIMENTS)
  if (scalar) {
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(stem_bwd_weight_k, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (const float*)g, (const float*)y, bn, x, dw, B, H, W, Ho, Wo);
  } else
#endif
  {
    // three workgroups fit a CU (44 KB of LDS each): 768 persistent workgroups are all resident - with 1 024 the last quarter only
    // started when the first finished, five tile times for 2.5 tiles per workgroup
    const int resident = 3 * 256;
    const int band_rows = stem_band_rows(B, Ho, resident, kWgBand);
    const int nbands = (Ho + band_rows - 1) / band_rows;
    grid = B * nbands < resident ? B * nbands : resident;
    const size_t sm = ((size_t)(2 * kWgBand + 3) * W + (kBlock / kWave) * (kWgChunk * kStemC + kStemC * 25)) * sizeof(float);
    TTK_REQUIRE(sm <= 64 * 1024, "stem_bwd_weight: image too wide for the LDS band (W=%d)", W);
    TTK_ACT_DISPATCH_STEM(act_bf16, hipLaunchKernelGGL((stem_wgrad_mfma_k<ActT, GradT>), dim3(grid), dim3(kBlock), sm, (hipStream_t)stream, (const GradT*)g,
                                                  (const ActT*)y, bn, x, dw, partial, B, H, W, Ho, Wo, nbands, band_rows));
    if (partial) launch_fold_partials(partial, grid, 25 * kStemC, dw, accumulate, (hipStream_t)stream);
  }
  TTK_LAUNCH_CHECK("stem_bwd_weight");
}

}  // extern "C"
