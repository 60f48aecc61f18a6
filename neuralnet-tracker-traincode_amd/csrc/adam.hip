// Fused global-norm gradient clipping + Adam over all parameter tensors in two launches, with no
// host synchronisation.  Replaces Lightning's clip_grad_norm_(max_norm=1.0, "norm") followed by
// torch.optim.Adam.step (scripts/train_poseestimator.py:147-167,442-445): ~190 tensors, 3.2 M floats.
//
// Work is cut into fixed-size chunks (multi-tensor apply): chunk c covers elements
// [chunk_offset[c], chunk_offset[c]+chunk_size) of tensor chunk_tensor[c].
//   pass 1: partial[c] = sum g^2 over the chunk
//   pass 2: every block re-adds the partials in a fixed order (fp64) -> total norm -> clip coefficient
//           -> Adam update of its chunk.  Deterministic.
#include "ttk_common.h"

namespace ttk {

struct AdamTables {
  const int64_t* ptrs;          // [ntensors][4]: param, grad, exp_avg, exp_avg_sq (device addresses)
  const int32_t* numel;         // [ntensors]
  const int32_t* group;         // [ntensors]: index into lr[] / wd[]
  const int32_t* chunk_tensor;  // [nchunks]
  const int32_t* chunk_offset;  // [nchunks]
  float* steps;                 // [ntensors]: updates applied to each tensor so far (torch.optim.Adam's per-parameter `step`)
};
struct AdamHyper {
  float lr[TTK_ADAM_MAX_GROUPS], wd[TTK_ADAM_MAX_GROUPS];
  float beta1, beta2, eps, max_norm, grad_scale;
};

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < kBlock / kWave; ++i) s += red[i];
  __syncthreads();
  return s;
}

__global__ void __launch_bounds__(kBlock) grad_sqnorm_k(AdamTables t, int chunk_size, float* __restrict__ partial) {
  __shared__ float red[kBlock / kWave];
  const int c = blockIdx.x;
  const int ti = t.chunk_tensor[c], off = t.chunk_offset[c];
  const float* g = reinterpret_cast<const float*>(t.ptrs[4 * ti + 1]);
  // device-resident per-tensor step counters (nothing about the step count is a launch argument: the call can sit in a
  // captured hipGraph): bumped here by the tensor's first chunk, read by clip_adam_k after the kernel boundary; a
  // tensor without a gradient this step is not counted, like torch.optim.Adam
  if (g && off == 0 && threadIdx.x == 0) t.steps[ti] += 1.f;
  const int n = min(chunk_size, t.numel[ti] - off);
  float acc = 0.f;
  if (g) {
    const float* gc = g + off;
    const int n4 = ((reinterpret_cast<uintptr_t>(gc) & 15) == 0) ? n / 4 : 0;  // 16-byte body, scalar tail
    for (int i = threadIdx.x; i < n4; i += kBlock) {
      const float4 v = ld4(gc + 4 * i);
      acc = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, acc))));
    }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += kBlock) {
      const float v = gc[i];
      acc = fmaf(v, v, acc);
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[c] = acc;
}

__global__ void __launch_bounds__(kBlock) clip_adam_k(AdamTables t, AdamHyper h, int chunk_size, int nchunks,
                                                       const float* __restrict__ partial, float* __restrict__ out_norm,
                                                       const float* __restrict__ hyper_dev) {
  __shared__ double dred[kBlock];
  // total gradient norm: fixed-order fp64 sum of the chunk partials (identical in every block)
  double acc = 0.0;
  for (int i = threadIdx.x; i < nchunks; i += kBlock) acc += (double)partial[i];
  dred[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) dred[threadIdx.x] += dred[threadIdx.x + s];
    __syncthreads();
  }
  // grad_scale: the gradients in memory are SUMS over data-parallel replicas; every use below sees grad_scale * g
  const float total = h.grad_scale * (float)sqrt(dred[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0 && out_norm) *out_norm = total;
  float coef = h.grad_scale;
  if (h.max_norm > 0.f) coef *= fminf(h.max_norm / (total + 1.0e-6f), 1.f);  // torch.nn.utils.clip_grad_norm_

  const int c = blockIdx.x;
  const int ti = t.chunk_tensor[c], off = t.chunk_offset[c];
  float* p = reinterpret_cast<float*>(t.ptrs[4 * ti + 0]);
  const float* g = reinterpret_cast<const float*>(t.ptrs[4 * ti + 1]);
  float* m = reinterpret_cast<float*>(t.ptrs[4 * ti + 2]);
  float* v = reinterpret_cast<float*>(t.ptrs[4 * ti + 3]);
  if (!g) return;  // parameter without gradient this step: untouched, like torch.optim.Adam
  const int n = min(chunk_size, t.numel[ti] - off);
  // hyper_dev: learning rates / weight decays live in device memory (same launch arguments every replay of a graph)
  const int gi = t.group[ti];
  const float lr = hyper_dev ? hyper_dev[TTK_ADAM_HYPER_LR + gi] : h.lr[gi], wd = hyper_dev ? hyper_dev[TTK_ADAM_HYPER_WD + gi] : h.wd[gi];
  const double tstep = (double)t.steps[ti];  // bias corrections as torch computes them (Python floats = fp64)
  const float bc1 = (float)(1.0 - pow((double)h.beta1, tstep)), bc2 = (float)(1.0 - pow((double)h.beta2, tstep));
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  auto update = [&](float gr, float& pv, float& mv, float& vv) {
    gr *= coef;
    if (wd != 0.f) gr = fmaf(wd, pv, gr);  // Adam's L2 form of weight_decay
    mv = fmaf(h.beta1, mv, (1.f - h.beta1) * gr);
    vv = fmaf(h.beta2, vv, (1.f - h.beta2) * gr * gr);
    const float denom = sqrtf(vv) * inv_sqrt_bc2 + h.eps;
    pv = pv - step_size * (mv / denom);
  };
  p += off; g += off; m += off; v += off;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const int n4 = aligned ? n / 4 : 0;  // 16-byte body (7 streams of 16 B per lane), scalar tail
  for (int i = threadIdx.x; i < n4; i += kBlock) {
    const float4 g4 = ld4(g + 4 * i);
    float4 p4 = ld4(p + 4 * i), m4 = ld4(m + 4 * i), v4 = ld4(v + 4 * i);
    update(g4.x, p4.x, m4.x, v4.x);
    update(g4.y, p4.y, m4.y, v4.y);
    update(g4.z, p4.z, m4.z, v4.z);
    update(g4.w, p4.w, m4.w, v4.w);
    st4(m + 4 * i, m4);
    st4(v + 4 * i, v4);
    st4(p + 4 * i, p4);
  }
  for (int k = 4 * n4 + threadIdx.x; k < n; k += kBlock) {
    float pv = p[k], mv = m[k], vv = v[k];
    update(g[k], pv, mv, vv);
    m[k] = mv;
    v[k] = vv;
    p[k] = pv;
  }
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_clip_adam(const int64_t* ptrs, const int32_t* numel, const int32_t* group, const int32_t* chunk_tensor,
                  const int32_t* chunk_offset, int nchunks, int chunk_size, const float* lr4, const float* wd4, float beta1,
                  float beta2, float eps, float max_norm, float grad_scale, float* steps, float* partial,
                  float* out_norm, const float* hyper_dev, ttk_stream_t stream) {
  TTK_REQUIRE(ptrs && numel && group && chunk_tensor && chunk_offset && lr4 && wd4 && partial && steps, "clip_adam: null pointer");
  TTK_REQUIRE(nchunks > 0 && chunk_size > 0, "clip_adam: bad chunking");
  TTK_REQUIRE(grad_scale > 0.f, "clip_adam: grad_scale must be positive (1 / number of replicas)");
  AdamTables t{ptrs, numel, group, chunk_tensor, chunk_offset, steps};
  AdamHyper h;
  for (int i = 0; i < TTK_ADAM_MAX_GROUPS; ++i) { h.lr[i] = lr4[i]; h.wd[i] = wd4[i]; }
  h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.max_norm = max_norm; h.grad_scale = grad_scale;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_sqnorm_k, dim3(nchunks), dim3(kBlock), 0, st, t, chunk_size, partial);
  hipLaunchKernelGGL(clip_adam_k, dim3(nchunks), dim3(kBlock), 0, st, t, h, chunk_size, nchunks, partial, out_norm, hyper_dev);
  TTK_LAUNCH_CHECK("clip_adam");
}

}  // extern "C"
