// On-GPU intensity augmentation of a batch of grey-level crops (reference call site
// trackertraincode/pipelines.py:508-532, container datatransformation/batch/intensity.py:30-41).  The reference runs
// seven kornia augmentations one after the other - each a full pass over the batch plus a per-sample select - followed by
// a clip.  Here ONE workgroup owns ONE image: it is read from HBM once, lives in LDS while every selected operation is
// applied in the reference's order, and is written once:
//
//   equalize  -> posterize -> gamma -> contrast -> brightness -> 5x5 Gaussian blur -> + noise -> clip -> + out_shift
//
// Per-sample parameters (which operations fire, and their sampled magnitudes) come from the host as a table
// prm[B][TTK_INTENSITY_PARAMS]; a disabled operation costs nothing.  The formulas restate kornia's published
// implementations (kornia.enhance.equalize / posterize / adjust_gamma / adjust_contrast / adjust_brightness,
// kornia.filters.gaussian_blur2d with border_type="reflect"); kornia itself is not available to this build: PARITY
// UNPINNED, the checker is oracle/intensity.py.
#include "ttk_common.h"

namespace ttk {

constexpr int kIntBlock = 256;

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ int reflect_idx(int i, int n) {  // "reflect": -1 -> 1, n -> n-2 (no edge repeat)
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

__global__ void __launch_bounds__(kIntBlock) intensity_augment_k(const float* __restrict__ x, float* __restrict__ y,
                                                                   const float* __restrict__ prm, const float* __restrict__ noise,
                                                                   int HW, int H, int W, float out_shift) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* img = sm;             // [HW]
  float* tmp = sm + HW;        // [HW] horizontal blur pass
  int* hist = reinterpret_cast<int*>(sm + 2 * HW);  // [256]
  float* lut = sm + 2 * HW + 256;                   // [256]
  __shared__ int s_step;
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* p = prm + (size_t)n * TTK_INTENSITY_PARAMS;
  const bool do_eq = p[TTK_INTENSITY_EQUALIZE] > 0.f;
  const int bits = (int)p[TTK_INTENSITY_POSTERIZE_BITS];
  const float gamma = p[TTK_INTENSITY_GAMMA], contrast = p[TTK_INTENSITY_CONTRAST], bright = p[TTK_INTENSITY_BRIGHTNESS];
  const bool do_blur = p[TTK_INTENSITY_BLUR] > 0.f;
  const float nstd = p[TTK_INTENSITY_NOISE_STD];
  const float* xi = x + (size_t)n * HW;

  if (do_eq) {
    hist[tid] = 0;  // kIntBlock == 256 bins
    __syncthreads();
  }
  for (int i = tid; i < HW; i += kIntBlock) {
    const float v = xi[i];
    img[i] = v;
    if (do_eq) {
      // torch.histc(im * 255, bins=256, min=0, max=255): bin = floor(v / 255 * 256), the maximum lands in the last
      // bin, values outside [0, 255] are ignored
      const float s = v * 255.f;
      if (s >= 0.f && s <= 255.f) atomicAdd(&hist[min((int)(s * (256.f / 255.f)), 255)], 1);
    }
  }
  __syncthreads();
  if (do_eq) {
    // step = (sum of the non-empty bins - the last non-empty bin) // 255;  lut[k] = (cumsum[k-1] + step // 2) // step,
    // lut[0] = 0, clamped to 0..255; step == 0 leaves the image alone
    if (tid == 0) {
      int total = 0, last = 0;
      for (int k = 0; k < 256; ++k) {
        total += hist[k];
        if (hist[k] != 0) last = hist[k];
      }
      s_step = (total - last) / 255;
    }
    __syncthreads();
    const int step = s_step;
    if (step > 0) {
      int cum = 0;
      for (int k = 0; k < tid; ++k) cum += hist[k];  // cumsum up to bin tid-1: 256 threads x <=255 adds from LDS
      lut[tid] = (float)min(max((cum + step / 2) / step, 0), 255) * (tid == 0 ? 0.f : 1.f);
    }
    __syncthreads();
    if (step > 0)
      for (int i = tid; i < HW; i += kIntBlock) {
        const int k = min(max((int)(img[i] * 255.f), 0), 255);  // im.long(): truncation
        img[i] = lut[k] / 255.f;
      }
  }
  // element-wise chain (each thread touches only its own pixels: no barrier needed in between)
  for (int i = tid; i < HW; i += kIntBlock) {
    float v = img[i];
    if (bits > 0 && bits < 8) {
      // posterize: uint8(v * 255) with the low 8-bits bits cleared
      const int u = min(max((int)(v * 255.f), 0), 255);
      v = (float)((u >> (8 - bits)) << (8 - bits)) / 255.f;
    }
    if (gamma > 0.f) v = clamp01(powf(v, gamma));      // adjust_gamma, gain 1
    if (contrast > 0.f) v = clamp01(v * contrast);     // adjust_contrast (multiplicative form)
    if (bright > 0.f) v = clamp01(v + (bright - 1.f));  // adjust_brightness(factor - 1)
    img[i] = v;
  }
  if (do_blur) {
    // separable 5-tap Gaussian, sigma 1.5, reflect border
    float g[5];
    float gs = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) { g[k] = expf(-(float)((k - 2) * (k - 2)) / (2.f * 1.5f * 1.5f)); gs += g[k]; }
#pragma unroll
    for (int k = 0; k < 5; ++k) g[k] /= gs;
    __syncthreads();
    for (int i = tid; i < HW; i += kIntBlock) {
      const int r = i / W, c = i - r * W;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) a = fmaf(g[k], img[r * W + reflect_idx(c + k - 2, W)], a);
      tmp[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < HW; i += kIntBlock) {
      const int r = i / W, c = i - r * W;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) a = fmaf(g[k], tmp[reflect_idx(r + k - 2, H) * W + c], a);
      img[i] = a;
    }
    // the vertical pass wrote only this thread's own pixels of img and read tmp: no barrier before the final loop
  }
  float* yo = y + (size_t)n * HW;
  const float* nz = (noise && nstd > 0.f) ? noise + (size_t)n * HW : nullptr;
  for (int i = tid; i < HW; i += kIntBlock) {
    float v = img[i];
    if (nz) v = fmaf(nstd, nz[i], v);
    yo[i] = clamp01(v) + out_shift;
  }
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_intensity_augment(const float* x, float* y, const float* params, const float* noise, int B, int H, int W, float out_shift,
                          ttk_stream_t stream) {
  TTK_REQUIRE(x && y && params && B > 0 && H >= 3 && W >= 3, "intensity_augment: bad arguments");
  const size_t sm = ((size_t)2 * H * W + 512) * sizeof(float);
  TTK_REQUIRE(sm <= 160 * 1024 - 64, "intensity_augment: image %dx%d does not fit in LDS (2*H*W floats + 2 KB <= 160 KB)", H, W);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(intensity_augment_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    TTK_REQUIRE(e == hipSuccess, "intensity_augment: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(intensity_augment_k, dim3(B), dim3(kIntBlock), sm, (hipStream_t)stream, x, y, params, noise, H * W, H, W, out_shift);
  TTK_LAUNCH_CHECK("intensity_augment");
}

}  // extern "C"
