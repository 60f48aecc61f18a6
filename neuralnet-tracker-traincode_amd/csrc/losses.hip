// Per-sample loss kernels (forward value + gradient w.r.t. the predictions), one thread per sample
// or per element; arithmetic in loss_math.h.  Reference: neuralnets/losses.py, negloglikelihood.py.
// All of these are a few KB per launch - they exist so that the loss path has no PyTorch fallback
// and no per-step host synchronisation (reference trackertraincode/train.py:433-438 stalls the
// stream every step; here values stay on the device).
#include "loss_math.h"
#include "ttk_common.h"

namespace ttk {

// The reductions over a sample's elements (68x3 landmarks, 50 shape parameters, ...) run one WAVE per
// sample: lanes stride over the elements (coalesced) and a shuffle tree folds them - a single thread looping
// over 204 strided loads took 140 us for 512 samples.
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off);
    hi = __shfl_xor(hi, off);
    v += __hiloint2double(hi, lo);
  }
  return v;
}
#define TTK_WAVE_SAMPLE(n)                                            \
  const int lane = threadIdx.x & 63;                                  \
  const int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  \
  if (s >= (n)) return;

#define TTK_SAMPLE_INDEX(n)                                         \
  const int s = blockIdx.x * blockDim.x + threadIdx.x;              \
  if (s >= (n)) return;

__device__ __forceinline__ void loss_rot_fwd_body(const float* q, const float* t, int n, float* v) {
  TTK_SAMPLE_INDEX(n);
  v[s] = lm::rot_loss(q + 4 * s, t + 4 * s);
}
__device__ __forceinline__ void loss_rot_bwd_body(const float* q, const float* t, const float* gv, int n, float* gq) {
  TTK_SAMPLE_INDEX(n);
  lm::rot_loss_bwd(q + 4 * s, t + 4 * s, gv[s], gq + 4 * s);
}
__device__ __forceinline__ void loss_rot6d_fwd_body(const float* R, const float* t, int n, float* v) {
  TTK_SAMPLE_INDEX(n);
  v[s] = lm::rot6d_loss(R + 9 * s, t + 4 * s);
}
__device__ __forceinline__ void loss_rot6d_bwd_body(const float* t, const float* gv, int n, float* gR) {
  TTK_SAMPLE_INDEX(n);
  lm::rot6d_loss_bwd(t + 4 * s, gv[s], gR + 9 * s);
}
__device__ __forceinline__ void loss_ortho6d_fwd_body(const float* z, int n, float* v) {
  TTK_SAMPLE_INDEX(n);
  v[s] = lm::ortho6d_loss(z + 6 * s);
}
__device__ __forceinline__ void loss_ortho6d_bwd_body(const float* z, const float* gv, int n, float* gz) {
  TTK_SAMPLE_INDEX(n);
  lm::ortho6d_loss_bwd(z + 6 * s, gv[s], gz + 6 * s);
}
__global__ void mat_to_quat_fwd_k(const float* m, int n, float* q) {
  TTK_SAMPLE_INDEX(n);
  lm::from_matrix(m + 9 * s, q + 4 * s);
}
__global__ void mat_to_quat_bwd_k(const float* m, const float* gq, int n, float* gm) {
  TTK_SAMPLE_INDEX(n);
  lm::from_matrix_bwd(m + 9 * s, gq + 4 * s, gm + 9 * s);
}
__device__ __forceinline__ void loss_quatreg_fwd_body(const float* q, int n, float* v) {
  TTK_SAMPLE_INDEX(n);
  v[s] = lm::quatreg_loss(q + 4 * s);
}
__device__ __forceinline__ void loss_quatreg_bwd_body(const float* q, const float* gv, int n, float* gq) {
  TTK_SAMPLE_INDEX(n);
  lm::quatreg_loss_bwd(q + 4 * s, gv[s], gq + 4 * s);
}
// mean_d (p - t)^2
__device__ __forceinline__ void loss_mse_rows_fwd_body(const float* p, const float* t, int n, int D, float* v) {
  TTK_WAVE_SAMPLE(n);
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float e = p[(size_t)s * D + d] - t[(size_t)s * D + d];
    acc = fmaf(e, e, acc);
  }
  acc = wave_sum_f(acc);
  if (lane == 0) v[s] = acc / (float)D;
}
__device__ __forceinline__ void loss_mse_rows_bwd_body(const float* p, const float* t, const float* gv, int n, int D, float* gp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * D) return;
  gp[i] = 2.f * (p[i] - t[i]) * gv[i / D] / (float)D;
}
// the same over the column window [c0, c0 + Dc) of rows that are Dt floats apart (p and t alike); the gradient is written
// for the whole row, zero outside the window
__device__ __forceinline__ void loss_mse_cols_fwd_body(const float* p, const float* t, int n, int Dt, int c0, int Dc, float* v) {
  TTK_WAVE_SAMPLE(n);
  float acc = 0.f;
  for (int d = lane; d < Dc; d += 64) {
    const float e = p[(size_t)s * Dt + c0 + d] - t[(size_t)s * Dt + c0 + d];
    acc = fmaf(e, e, acc);
  }
  acc = wave_sum_f(acc);
  if (lane == 0) v[s] = acc / (float)Dc;
}
__device__ __forceinline__ void loss_mse_cols_bwd_body(const float* p, const float* t, const float* gv, int n, int Dt, int c0, int Dc, float* gp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * Dt) return;
  const int d = i % Dt - c0;
  gp[i] = (d >= 0 && d < Dc) ? 2.f * (p[i] - t[i]) * gv[i / Dt] / (float)Dc : 0.f;
}

// The non-default kinds of the reference's loss switches (losses.py:16-38): v[s] = sum_d colw[d] * f_kind(p[s][d] - t[s][d]) over rows of
// D floats.  The host folds the reduction of each loss class into colw (mean over a column window: 1/Dc inside, 0 outside; landmarks:
// point weight / 68 on the first `dim` coordinates of each point).
__device__ __forceinline__ void loss_elem_fwd_body(const float* p, const float* t, const float* colw, int n, int D, int kind, float beta, float* v) {
  TTK_WAVE_SAMPLE(n);
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float w = colw[d];
    if (w != 0.f) acc = fmaf(w, lm::elem_loss(kind, p[(size_t)s * D + d] - t[(size_t)s * D + d], beta), acc);
  }
  acc = wave_sum_f(acc);
  if (lane == 0) v[s] = acc;
}
__device__ __forceinline__ void loss_elem_bwd_body(const float* p, const float* t, const float* colw, const float* gv, int n, int D, int kind, float beta,
                                                   float* gp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * D) return;
  const float w = colw[i % D];
  gp[i] = w != 0.f ? w * lm::elem_loss_d(kind, p[i] - t[i], beta) * gv[i / D] : 0.f;
}
__device__ __forceinline__ void loss_rot_geodesic_fwd_body(const float* q, const float* t, int n, float* v) {
  TTK_SAMPLE_INDEX(n);
  v[s] = lm::smooth_geodesic_loss(q + 4 * s, t + 4 * s);
}
__device__ __forceinline__ void loss_rot_geodesic_bwd_body(const float* q, const float* t, const float* gv, int n, float* gq) {
  TTK_SAMPLE_INDEX(n);
  lm::smooth_geodesic_loss_bwd(q + 4 * s, t + 4 * s, gv[s], gq + 4 * s);
}

// ---- loss bookkeeping of one training step in single launches (train.py default_compute_loss) ---------------------
constexpr int kMaxSeg = 32;
struct CopySegs {
  const float* src[kMaxSeg];  // nullptr: zeros
  float* dst[kMaxSeg];
  long long first[kMaxSeg + 1];  // running element offsets
  int n;
};
__global__ void multi_copy_k(CopySegs a) {
  const long long total = a.first[a.n];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int k = 0;
    while (i >= a.first[k + 1]) ++k;
    const long long j = i - a.first[k];
    a.dst[k][j] = a.src[k] ? a.src[k][j] : 0.f;
  }
}
struct SumTerms {
  const float* val[kMaxSeg];   // forward: loss values; backward: unused
  float* gval[kMaxSeg];        // backward: gradient of each term's values
  const float* sw[kMaxSeg];    // per-sample weights or nullptr (= 1)
  float w[kMaxSeg];
  int first[kMaxSeg + 1];
  int n;
};
// out = scale * sum_k w_k * sum_i sw_k[i] * val_k[i]; one block, fixed summation order (double accumulators)
__global__ void __launch_bounds__(1024) weighted_sum_fwd_k(SumTerms a, float scale, float* out) {
  __shared__ double red[1024];
  double acc = 0.0;
  const int total = a.first[a.n];
  for (int i = threadIdx.x; i < total; i += 1024) {
    int k = 0;
    while (i >= a.first[k + 1]) ++k;
    const int j = i - a.first[k];
    acc += (double)(a.val[k][j] * (a.sw[k] ? a.w[k] * a.sw[k][j] : a.w[k]));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)(red[0] * (double)scale);
}
__global__ void weighted_sum_bwd_k(SumTerms a, float scale, const float* gout) {
  const int total = a.first[a.n];
  const float g = *gout * scale;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int k = 0;
    while (i >= a.first[k + 1]) ++k;
    const int j = i - a.first[k];
    a.gval[k][j] = g * (a.sw[k] ? a.w[k] * a.sw[k][j] : a.w[k]);
  }
}
// mean_p( w_p * sum_{d<dim} (p - t)^2 )
__device__ __forceinline__ void loss_points_fwd_body(const float* p, const float* t, int n, int dim, float chin, float eye, float* v) {
  TTK_WAVE_SAMPLE(n);
  float acc = 0.f;
  for (int k = lane; k < 68; k += 64) {
    float e2 = 0.f;
    for (int d = 0; d < dim; ++d) {
      const float e = p[((size_t)s * 68 + k) * 3 + d] - t[((size_t)s * 68 + k) * 3 + d];
      e2 = fmaf(e, e, e2);
    }
    acc = fmaf(lm::point_weight(k, chin, eye), e2, acc);
  }
  acc = wave_sum_f(acc);
  if (lane == 0) v[s] = acc / 68.f;
}
__device__ __forceinline__ void loss_points_bwd_body(const float* p, const float* t, const float* gv, int n, int dim, float chin, float eye,
                                  float* gp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 204) return;
  const int s = i / 204, r = i % 204, k = r / 3, d = r % 3;
  gp[i] = d < dim ? 2.f * lm::point_weight(k, chin, eye) * (p[i] - t[i]) * gv[s] / 68.f : 0.f;
}
__device__ __forceinline__ void loss_nllrot_fwd_body(const float* q, const float* t, const float* L, int n, float* v) {
  TTK_SAMPLE_INDEX(n);
  v[s] = lm::nllrot_loss(q + 4 * s, t + 4 * s, L + 9 * s);
}
__device__ __forceinline__ void loss_nllrot_bwd_body(const float* q, const float* t, const float* L, const float* gv, int n, float* gq, float* gL) {
  TTK_SAMPLE_INDEX(n);
  lm::nllrot_loss_bwd(q + 4 * s, t + 4 * s, L + 9 * s, gv[s], gq + 4 * s, gL + 9 * s);
}
__device__ __forceinline__ void loss_nllcoord_fwd_body(const float* c, const float* t, const float* L, int n, float* v) {
  TTK_SAMPLE_INDEX(n);
  v[s] = lm::nllcoord_loss(c + 3 * s, t + 3 * s, L + 9 * s);
}
__device__ __forceinline__ void loss_nllcoord_bwd_body(const float* c, const float* t, const float* L, const float* gv, int n, float* gc, float* gL) {
  TTK_SAMPLE_INDEX(n);
  lm::nllcoord_loss_bwd(c + 3 * s, t + 3 * s, L + 9 * s, gv[s], gc + 3 * s, gL + 9 * s);
}
// -mean over `per` elements of w * Normal(mu, sigma).log_prob(x); elements laid out [n][rows][3] with
// only the first `dim` of every 3 used when rows3 != 0 (points), else plain [n][per].
template <bool LAPLACE = false>
__device__ __forceinline__ void loss_normal_fwd_body(const float* mu, const float* sg, const float* x, int n, int per, int points, int dim,
                                  float chin, float eye, float* v) {
  TTK_WAVE_SAMPLE(n);
  float acc = 0.f;
  if (points) {
    for (int k = lane; k < 68; k += 64) {
      float a = 0.f;
      for (int d = 0; d < dim; ++d) {
        const size_t o = ((size_t)s * 68 + k) * 3 + d;
        a += LAPLACE ? lm::laplace_nll(mu[o], sg[o], x[o]) : lm::normal_nll(mu[o], sg[o], x[o]);
      }
      acc = fmaf(lm::point_weight(k, chin, eye), a, acc);
    }
    acc = wave_sum_f(acc);
    if (lane == 0) v[s] = acc / (68.f * (float)dim);
  } else {
    for (int d = lane; d < per; d += 64) {
      const size_t o = (size_t)s * per + d;
      acc += LAPLACE ? lm::laplace_nll(mu[o], sg[o], x[o]) : lm::normal_nll(mu[o], sg[o], x[o]);
    }
    acc = wave_sum_f(acc);
    if (lane == 0) v[s] = acc / (float)per;
  }
}
template <bool LAPLACE = false>
__device__ __forceinline__ void loss_normal_bwd_body(const float* mu, const float* sg, const float* x, const float* gv, int n, int per,
                                  int points, int dim, float chin, float eye, float* gmu, float* gsg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = points ? 204 : per;
  if (i >= n * stride) return;
  const int s = i / stride, r = i % stride;
  float w;
  if (points) {
    const int k = r / 3, d = r % 3;
    w = d < dim ? lm::point_weight(k, chin, eye) / (68.f * (float)dim) : 0.f;
  } else {
    w = 1.f / (float)per;
  }
  float a = 0.f, b = 0.f;
  if (w != 0.f) {
    if (LAPLACE) lm::laplace_nll_bwd(mu[i], sg[i], x[i], w * gv[s], a, b);
    else lm::normal_nll_bwd(mu[i], sg[i], x[i], w * gv[s], a, b);
  }
  gmu[i] = a;
  gsg[i] = b;
}
__device__ __forceinline__ void loss_gmm_fwd_body(const float* x, const double* ck, const double* mu, const double* sinv, int K, double fudge,
                               int n, float* v, double* post) {
  TTK_WAVE_SAMPLE(n);
  double a[16];
  const double xd = lane < 50 ? (double)x[50 * s + lane] : 0.0;
  double mx = -1.0e300;
  for (int k = 0; k < K; ++k) {
    double z = 0.0;
    if (lane < 50) {
      z = (xd - mu[k * 50 + lane]) * sinv[k * 50 + lane];
      z *= z;
    }
    a[k] = ck[k] - 0.5 * wave_sum_d(z);  // same value in every lane
    mx = a[k] > mx ? a[k] : mx;
  }
  double sum = 0.0;
  for (int k = 0; k < K; ++k) sum += exp(a[k] - mx);
  for (int k = 0; k < K; ++k)  // static index: a[] stays in registers
    if (lane == k) post[(size_t)K * s + k] = exp(a[k] - mx) / sum;
  if (lane == 0) v[s] = (float)(-(mx + log(sum)) * fudge);
}
// d/dx_d = fudge * sum_k post_k (x_d - mu_kd) sinv_kd^2
__device__ __forceinline__ void loss_gmm_bwd_body(const float* x, const double* mu, const double* sinv, const double* post, int K, double fudge,
                               const float* gv, int n, float* gx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 50) return;
  const int s = i / 50, d = i % 50;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) {
    const double si = sinv[k * 50 + d];
    acc += post[(size_t)K * s + k] * ((double)x[i] - mu[k * 50 + d]) * si * si;
  }
  gx[i] = (float)(fudge * acc * (double)gv[s]);
}

// ---- the thin kernels of the single-op entry points ------------------------------------------------------------------
__global__ void loss_rot_fwd_k(const float* q, const float* t, int n, float* v) { loss_rot_fwd_body(q, t, n, v); }
__global__ void loss_rot_bwd_k(const float* q, const float* t, const float* gv, int n, float* gq) { loss_rot_bwd_body(q, t, gv, n, gq); }
__global__ void loss_rot6d_fwd_k(const float* R, const float* t, int n, float* v) { loss_rot6d_fwd_body(R, t, n, v); }
__global__ void loss_rot6d_bwd_k(const float* t, const float* gv, int n, float* gR) { loss_rot6d_bwd_body(t, gv, n, gR); }
__global__ void loss_ortho6d_fwd_k(const float* z, int n, float* v) { loss_ortho6d_fwd_body(z, n, v); }
__global__ void loss_ortho6d_bwd_k(const float* z, const float* gv, int n, float* gz) { loss_ortho6d_bwd_body(z, gv, n, gz); }
__global__ void loss_quatreg_fwd_k(const float* q, int n, float* v) { loss_quatreg_fwd_body(q, n, v); }
__global__ void loss_quatreg_bwd_k(const float* q, const float* gv, int n, float* gq) { loss_quatreg_bwd_body(q, gv, n, gq); }
__global__ void loss_mse_rows_fwd_k(const float* p, const float* t, int n, int D, float* v) { loss_mse_rows_fwd_body(p, t, n, D, v); }
__global__ void loss_mse_rows_bwd_k(const float* p, const float* t, const float* gv, int n, int D, float* gp) { loss_mse_rows_bwd_body(p, t, gv, n, D, gp); }
__global__ void loss_mse_cols_fwd_k(const float* p, const float* t, int n, int Dt, int c0, int Dc, float* v) { loss_mse_cols_fwd_body(p, t, n, Dt, c0, Dc, v); }
__global__ void loss_mse_cols_bwd_k(const float* p, const float* t, const float* gv, int n, int Dt, int c0, int Dc, float* gp) { loss_mse_cols_bwd_body(p, t, gv, n, Dt, c0, Dc, gp); }
__global__ void loss_points_fwd_k(const float* p, const float* t, int n, int dim, float chin, float eye, float* v) { loss_points_fwd_body(p, t, n, dim, chin, eye, v); }
__global__ void loss_points_bwd_k(const float* p, const float* t, const float* gv, int n, int dim, float chin, float eye, float* gp) { loss_points_bwd_body(p, t, gv, n, dim, chin, eye, gp); }
__global__ void loss_nllrot_fwd_k(const float* q, const float* t, const float* L, int n, float* v) { loss_nllrot_fwd_body(q, t, L, n, v); }
__global__ void loss_nllrot_bwd_k(const float* q, const float* t, const float* L, const float* gv, int n, float* gq, float* gL) { loss_nllrot_bwd_body(q, t, L, gv, n, gq, gL); }
__global__ void loss_nllcoord_fwd_k(const float* c, const float* t, const float* L, int n, float* v) { loss_nllcoord_fwd_body(c, t, L, n, v); }
__global__ void loss_nllcoord_bwd_k(const float* c, const float* t, const float* L, const float* gv, int n, float* gc, float* gL) { loss_nllcoord_bwd_body(c, t, L, gv, n, gc, gL); }
__global__ void loss_normal_fwd_k(const float* mu, const float* sg, const float* x, int n, int per, int points, int dim, float chin, float eye, float* v) { loss_normal_fwd_body(mu, sg, x, n, per, points, dim, chin, eye, v); }
__global__ void loss_normal_bwd_k(const float* mu, const float* sg, const float* x, const float* gv, int n, int per, int points, int dim, float chin, float eye, float* gmu, float* gsg) { loss_normal_bwd_body(mu, sg, x, gv, n, per, points, dim, chin, eye, gmu, gsg); }
__global__ void loss_laplace_fwd_k(const float* mu, const float* sg, const float* x, int n, int per, int points, int dim, float chin, float eye, float* v) { loss_normal_fwd_body<true>(mu, sg, x, n, per, points, dim, chin, eye, v); }
__global__ void loss_laplace_bwd_k(const float* mu, const float* sg, const float* x, const float* gv, int n, int per, int points, int dim, float chin, float eye, float* gmu, float* gsg) { loss_normal_bwd_body<true>(mu, sg, x, gv, n, per, points, dim, chin, eye, gmu, gsg); }
__global__ void loss_elem_fwd_k(const float* p, const float* t, const float* colw, int n, int D, int kind, float beta, float* v) { loss_elem_fwd_body(p, t, colw, n, D, kind, beta, v); }
__global__ void loss_elem_bwd_k(const float* p, const float* t, const float* colw, const float* gv, int n, int D, int kind, float beta, float* gp) { loss_elem_bwd_body(p, t, colw, gv, n, D, kind, beta, gp); }
__global__ void loss_rot_geodesic_fwd_k(const float* q, const float* t, int n, float* v) { loss_rot_geodesic_fwd_body(q, t, n, v); }
__global__ void loss_rot_geodesic_bwd_k(const float* q, const float* t, const float* gv, int n, float* gq) { loss_rot_geodesic_bwd_body(q, t, gv, n, gq); }
__global__ void loss_gmm_fwd_k(const float* x, const double* ck, const double* mu, const double* sinv, int K, double fudge, int n, float* v, double* post) { loss_gmm_fwd_body(x, ck, mu, sinv, K, fudge, n, v, post); }
__global__ void loss_gmm_bwd_k(const float* x, const double* mu, const double* sinv, const double* post, int K, double fudge, const float* gv, int n, float* gx) { loss_gmm_bwd_body(x, mu, sinv, post, K, fudge, gv, n, gx); }

// ---- many loss ops in ONE launch (ttk_loss_batch): blockIdx.y picks the op, blockIdx.x its 256-thread block --------------
// The ~30 loss kernels of a step are a few KB each and independent of one another, but on one stream each costs its ~4.5 us of
// launch-to-completion latency; batched they cost one.  Arguments sit in the op in the order of the entry point's signature:
// pointers in p[], ints in i[], floats in f[], the double in d.
constexpr int kBatchMax = TTK_LOSS_BATCH_MAX;
struct BatchArgs {
  ttk_loss_op op[kBatchMax];
};
#define P_(k) static_cast<const float*>(o.p[k])
#define W_(k) static_cast<float*>(const_cast<void*>(o.p[k]))
#define D_(k) static_cast<const double*>(o.p[k])
__global__ void __launch_bounds__(256) loss_batch_k(BatchArgs a) {
  const ttk_loss_op& o = a.op[blockIdx.y];
  if ((long long)blockIdx.x * 256 >= (long long)o.items) return;
  switch (o.kind) {
    case TTK_OP_ROT_FWD: loss_rot_fwd_body(P_(0), P_(1), o.i[0], W_(2)); break;
    case TTK_OP_ROT_BWD: loss_rot_bwd_body(P_(0), P_(1), P_(2), o.i[0], W_(3)); break;
    case TTK_OP_ROT6D_FWD: loss_rot6d_fwd_body(P_(0), P_(1), o.i[0], W_(2)); break;
    case TTK_OP_ROT6D_BWD: loss_rot6d_bwd_body(P_(0), P_(1), o.i[0], W_(2)); break;
    case TTK_OP_ORTHO6D_FWD: loss_ortho6d_fwd_body(P_(0), o.i[0], W_(1)); break;
    case TTK_OP_ORTHO6D_BWD: loss_ortho6d_bwd_body(P_(0), P_(1), o.i[0], W_(2)); break;
    case TTK_OP_QUATREG_FWD: loss_quatreg_fwd_body(P_(0), o.i[0], W_(1)); break;
    case TTK_OP_QUATREG_BWD: loss_quatreg_bwd_body(P_(0), P_(1), o.i[0], W_(2)); break;
    case TTK_OP_MSE_ROWS_FWD: loss_mse_rows_fwd_body(P_(0), P_(1), o.i[0], o.i[1], W_(2)); break;
    case TTK_OP_MSE_ROWS_BWD: loss_mse_rows_bwd_body(P_(0), P_(1), P_(2), o.i[0], o.i[1], W_(3)); break;
    case TTK_OP_MSE_COLS_FWD: loss_mse_cols_fwd_body(P_(0), P_(1), o.i[0], o.i[1], o.i[2], o.i[3], W_(2)); break;
    case TTK_OP_MSE_COLS_BWD: loss_mse_cols_bwd_body(P_(0), P_(1), P_(2), o.i[0], o.i[1], o.i[2], o.i[3], W_(3)); break;
    case TTK_OP_POINTS_FWD: loss_points_fwd_body(P_(0), P_(1), o.i[0], o.i[1], o.f[0], o.f[1], W_(2)); break;
    case TTK_OP_POINTS_BWD: loss_points_bwd_body(P_(0), P_(1), P_(2), o.i[0], o.i[1], o.f[0], o.f[1], W_(3)); break;
    case TTK_OP_NLLROT_FWD: loss_nllrot_fwd_body(P_(0), P_(1), P_(2), o.i[0], W_(3)); break;
    case TTK_OP_NLLROT_BWD: loss_nllrot_bwd_body(P_(0), P_(1), P_(2), P_(3), o.i[0], W_(4), W_(5)); break;
    case TTK_OP_NLLCOORD_FWD: loss_nllcoord_fwd_body(P_(0), P_(1), P_(2), o.i[0], W_(3)); break;
    case TTK_OP_NLLCOORD_BWD: loss_nllcoord_bwd_body(P_(0), P_(1), P_(2), P_(3), o.i[0], W_(4), W_(5)); break;
    case TTK_OP_NORMAL_FWD: loss_normal_fwd_body(P_(0), P_(1), P_(2), o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.f[1], W_(3)); break;
    case TTK_OP_NORMAL_BWD: loss_normal_bwd_body(P_(0), P_(1), P_(2), P_(3), o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.f[1], W_(4), W_(5)); break;
    case TTK_OP_GMM_FWD: loss_gmm_fwd_body(P_(0), D_(1), D_(2), D_(3), o.i[0], o.d, o.i[1], W_(4), static_cast<double*>(const_cast<void*>(o.p[5]))); break;
    case TTK_OP_GMM_BWD: loss_gmm_bwd_body(P_(0), D_(1), D_(2), D_(3), o.i[0], o.d, P_(4), o.i[1], W_(5)); break;
    default: break;
  }
}
#undef P_
#undef W_
#undef D_

}  // namespace ttk


using namespace ttk;

#define TTK_GRID(n) dim3(((n) + 255) / 256), dim3(256), 0, (hipStream_t)stream

extern "C" {

int ttk_loss_rot_fwd(const float* q, const float* t, int n, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(q && t && v && n > 0, "loss_rot_fwd: bad arguments");
  hipLaunchKernelGGL(loss_rot_fwd_k, TTK_GRID(n), q, t, n, v);
  TTK_LAUNCH_CHECK("loss_rot_fwd");
}
int ttk_loss_rot_bwd(const float* q, const float* t, const float* gv, int n, float* gq, ttk_stream_t stream) {
  TTK_REQUIRE(q && t && gv && gq && n > 0, "loss_rot_bwd: bad arguments");
  hipLaunchKernelGGL(loss_rot_bwd_k, TTK_GRID(n), q, t, gv, n, gq);
  TTK_LAUNCH_CHECK("loss_rot_bwd");
}
int ttk_loss_rot6d_fwd(const float* R, const float* t, int n, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(R && t && v && n > 0, "loss_rot6d_fwd: bad arguments");
  hipLaunchKernelGGL(loss_rot6d_fwd_k, TTK_GRID(n), R, t, n, v);
  TTK_LAUNCH_CHECK("loss_rot6d_fwd");
}
int ttk_loss_rot6d_bwd(const float* t, const float* gv, int n, float* gR, ttk_stream_t stream) {
  TTK_REQUIRE(t && gv && gR && n > 0, "loss_rot6d_bwd: bad arguments");
  hipLaunchKernelGGL(loss_rot6d_bwd_k, TTK_GRID(n), t, gv, n, gR);
  TTK_LAUNCH_CHECK("loss_rot6d_bwd");
}
int ttk_loss_ortho6d_fwd(const float* z, int n, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(z && v && n > 0, "loss_ortho6d_fwd: bad arguments");
  hipLaunchKernelGGL(loss_ortho6d_fwd_k, TTK_GRID(n), z, n, v);
  TTK_LAUNCH_CHECK("loss_ortho6d_fwd");
}
int ttk_loss_ortho6d_bwd(const float* z, const float* gv, int n, float* gz, ttk_stream_t stream) {
  TTK_REQUIRE(z && gv && gz && n > 0, "loss_ortho6d_bwd: bad arguments");
  hipLaunchKernelGGL(loss_ortho6d_bwd_k, TTK_GRID(n), z, gv, n, gz);
  TTK_LAUNCH_CHECK("loss_ortho6d_bwd");
}
int ttk_mat_to_quat_fwd(const float* m, int n, float* q, ttk_stream_t stream) {
  TTK_REQUIRE(m && q && n > 0, "mat_to_quat_fwd: bad arguments");
  hipLaunchKernelGGL(mat_to_quat_fwd_k, TTK_GRID(n), m, n, q);
  TTK_LAUNCH_CHECK("mat_to_quat_fwd");
}
int ttk_mat_to_quat_bwd(const float* m, const float* gq, int n, float* gm, ttk_stream_t stream) {
  TTK_REQUIRE(m && gq && gm && n > 0, "mat_to_quat_bwd: bad arguments");
  hipLaunchKernelGGL(mat_to_quat_bwd_k, TTK_GRID(n), m, gq, n, gm);
  TTK_LAUNCH_CHECK("mat_to_quat_bwd");
}
int ttk_loss_quatreg_fwd(const float* q, int n, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(q && v && n > 0, "loss_quatreg_fwd: bad arguments");
  hipLaunchKernelGGL(loss_quatreg_fwd_k, TTK_GRID(n), q, n, v);
  TTK_LAUNCH_CHECK("loss_quatreg_fwd");
}
int ttk_loss_quatreg_bwd(const float* q, const float* gv, int n, float* gq, ttk_stream_t stream) {
  TTK_REQUIRE(q && gv && gq && n > 0, "loss_quatreg_bwd: bad arguments");
  hipLaunchKernelGGL(loss_quatreg_bwd_k, TTK_GRID(n), q, gv, n, gq);
  TTK_LAUNCH_CHECK("loss_quatreg_bwd");
}
int ttk_loss_mse_rows_fwd(const float* p, const float* t, int n, int D, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(p && t && v && n > 0 && D > 0, "loss_mse_rows_fwd: bad arguments");
  hipLaunchKernelGGL(loss_mse_rows_fwd_k, TTK_GRID(n * 64), p, t, n, D, v);
  TTK_LAUNCH_CHECK("loss_mse_rows_fwd");
}
int ttk_loss_mse_rows_bwd(const float* p, const float* t, const float* gv, int n, int D, float* gp, ttk_stream_t stream) {
  TTK_REQUIRE(p && t && gv && gp && n > 0 && D > 0, "loss_mse_rows_bwd: bad arguments");
  hipLaunchKernelGGL(loss_mse_rows_bwd_k, TTK_GRID(n * D), p, t, gv, n, D, gp);
  TTK_LAUNCH_CHECK("loss_mse_rows_bwd");
}
int ttk_loss_mse_cols_fwd(const float* p, const float* t, int n, int Dt, int c0, int Dc, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(p && t && v && n > 0 && Dc > 0 && c0 >= 0 && c0 + Dc <= Dt, "loss_mse_cols_fwd: bad arguments");
  hipLaunchKernelGGL(loss_mse_cols_fwd_k, TTK_GRID(n * 64), p, t, n, Dt, c0, Dc, v);
  TTK_LAUNCH_CHECK("loss_mse_cols_fwd");
}
int ttk_loss_mse_cols_bwd(const float* p, const float* t, const float* gv, int n, int Dt, int c0, int Dc, float* gp,
                          ttk_stream_t stream) {
  TTK_REQUIRE(p && t && gv && gp && n > 0 && Dc > 0 && c0 >= 0 && c0 + Dc <= Dt, "loss_mse_cols_bwd: bad arguments");
  hipLaunchKernelGGL(loss_mse_cols_bwd_k, TTK_GRID(n * Dt), p, t, gv, n, Dt, c0, Dc, gp);
  TTK_LAUNCH_CHECK("loss_mse_cols_bwd");
}
int ttk_multi_copy(int n, const float* const* src, float* const* dst, const int64_t* count, ttk_stream_t stream) {
  TTK_REQUIRE(n > 0 && n <= kMaxSeg && src && dst && count, "multi_copy: 1..32 segments");
  CopySegs a{};
  a.n = n;
  long long total = 0;
  for (int k = 0; k < n; ++k) {
    TTK_REQUIRE(count[k] >= 0 && (dst[k] || count[k] == 0), "multi_copy: segment %d: null destination or negative count", k);
    a.src[k] = src[k];
    a.dst[k] = dst[k];
    a.first[k] = total;
    total += count[k];
  }
  a.first[n] = total;
  if (total == 0) return 0;
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(multi_copy_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  TTK_LAUNCH_CHECK("multi_copy");
}
static int fill_sum_terms(SumTerms& a, int n, const float* const* sample_w, const float* w, const int* count) {
  a.n = n;
  int total = 0;
  for (int k = 0; k < n; ++k) {
    if (count[k] < 0) return -1;
    a.sw[k] = sample_w ? sample_w[k] : nullptr;
    a.w[k] = w[k];
    a.first[k] = total;
    total += count[k];
  }
  a.first[n] = total;
  return total;
}
int ttk_weighted_sum_fwd(int n, const float* const* val, const float* const* sample_w, const float* w, const int* count,
                         float scale, float* out, ttk_stream_t stream) {
  TTK_REQUIRE(n > 0 && n <= kMaxSeg && val && w && count && out, "weighted_sum_fwd: 1..32 terms");
  SumTerms a{};
  TTK_REQUIRE(fill_sum_terms(a, n, sample_w, w, count) >= 0, "weighted_sum_fwd: negative count");
  for (int k = 0; k < n; ++k) {
    TTK_REQUIRE(val[k] || count[k] == 0, "weighted_sum_fwd: term %d: null values", k);
    a.val[k] = val[k];
  }
  hipLaunchKernelGGL(weighted_sum_fwd_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, scale, out);
  TTK_LAUNCH_CHECK("weighted_sum_fwd");
}
int ttk_weighted_sum_bwd(int n, const float* gout, const float* const* sample_w, const float* w, const int* count, float scale,
                         float* const* gval, ttk_stream_t stream) {
  TTK_REQUIRE(n > 0 && n <= kMaxSeg && gout && w && count && gval, "weighted_sum_bwd: 1..32 terms");
  SumTerms a{};
  const int total = fill_sum_terms(a, n, sample_w, w, count);
  TTK_REQUIRE(total >= 0, "weighted_sum_bwd: negative count");
  for (int k = 0; k < n; ++k) {
    TTK_REQUIRE(gval[k] || count[k] == 0, "weighted_sum_bwd: term %d: null gradient buffer", k);
    a.gval[k] = gval[k];
  }
  if (total == 0) return 0;
  hipLaunchKernelGGL(weighted_sum_bwd_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, scale, gout);
  TTK_LAUNCH_CHECK("weighted_sum_bwd");
}
int ttk_loss_points_fwd(const float* p, const float* t, int n, int dim, float chin, float eye, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(p && t && v && n > 0 && (dim == 2 || dim == 3), "loss_points_fwd: bad arguments");
  hipLaunchKernelGGL(loss_points_fwd_k, TTK_GRID(n * 64), p, t, n, dim, chin, eye, v);
  TTK_LAUNCH_CHECK("loss_points_fwd");
}
int ttk_loss_points_bwd(const float* p, const float* t, const float* gv, int n, int dim, float chin, float eye, float* gp,
                        ttk_stream_t stream) {
  TTK_REQUIRE(p && t && gv && gp && n > 0 && (dim == 2 || dim == 3), "loss_points_bwd: bad arguments");
  hipLaunchKernelGGL(loss_points_bwd_k, TTK_GRID(n * 204), p, t, gv, n, dim, chin, eye, gp);
  TTK_LAUNCH_CHECK("loss_points_bwd");
}
int ttk_loss_nllrot_fwd(const float* q, const float* t, const float* L, int n, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(q && t && L && v && n > 0, "loss_nllrot_fwd: bad arguments");
  hipLaunchKernelGGL(loss_nllrot_fwd_k, TTK_GRID(n), q, t, L, n, v);
  TTK_LAUNCH_CHECK("loss_nllrot_fwd");
}
int ttk_loss_nllrot_bwd(const float* q, const float* t, const float* L, const float* gv, int n, float* gq, float* gL,
                        ttk_stream_t stream) {
  TTK_REQUIRE(q && t && L && gv && gq && gL && n > 0, "loss_nllrot_bwd: bad arguments");
  hipLaunchKernelGGL(loss_nllrot_bwd_k, TTK_GRID(n), q, t, L, gv, n, gq, gL);
  TTK_LAUNCH_CHECK("loss_nllrot_bwd");
}
int ttk_loss_nllcoord_fwd(const float* c, const float* t, const float* L, int n, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(c && t && L && v && n > 0, "loss_nllcoord_fwd: bad arguments");
  hipLaunchKernelGGL(loss_nllcoord_fwd_k, TTK_GRID(n), c, t, L, n, v);
  TTK_LAUNCH_CHECK("loss_nllcoord_fwd");
}
int ttk_loss_nllcoord_bwd(const float* c, const float* t, const float* L, const float* gv, int n, float* gc, float* gL,
                          ttk_stream_t stream) {
  TTK_REQUIRE(c && t && L && gv && gc && gL && n > 0, "loss_nllcoord_bwd: bad arguments");
  hipLaunchKernelGGL(loss_nllcoord_bwd_k, TTK_GRID(n), c, t, L, gv, n, gc, gL);
  TTK_LAUNCH_CHECK("loss_nllcoord_bwd");
}
int ttk_loss_normal_fwd(const float* mu, const float* sigma, const float* x, int n, int per, int points, int dim, float chin,
                        float eye, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(mu && sigma && x && v && n > 0 && (points ? (dim == 2 || dim == 3) : per > 0), "loss_normal_fwd: bad arguments");
  hipLaunchKernelGGL(loss_normal_fwd_k, TTK_GRID(n * 64), mu, sigma, x, n, per, points, dim, chin, eye, v);
  TTK_LAUNCH_CHECK("loss_normal_fwd");
}
int ttk_loss_normal_bwd(const float* mu, const float* sigma, const float* x, const float* gv, int n, int per, int points,
                        int dim, float chin, float eye, float* gmu, float* gsigma, ttk_stream_t stream) {
  TTK_REQUIRE(mu && sigma && x && gv && gmu && gsigma && n > 0, "loss_normal_bwd: bad arguments");
  const int total = n * (points ? 204 : per);
  hipLaunchKernelGGL(loss_normal_bwd_k, TTK_GRID(total), mu, sigma, x, gv, n, per, points, dim, chin, eye, gmu, gsigma);
  TTK_LAUNCH_CHECK("loss_normal_bwd");
}
int ttk_loss_laplace_fwd(const float* mu, const float* b, const float* x, int n, int per, int points, int dim, float chin, float eye,
                         float* v, ttk_stream_t stream) {
  TTK_REQUIRE(mu && b && x && v && n > 0 && (points ? (dim == 2 || dim == 3) : per > 0), "loss_laplace_fwd: bad arguments");
  hipLaunchKernelGGL(loss_laplace_fwd_k, TTK_GRID(n * 64), mu, b, x, n, per, points, dim, chin, eye, v);
  TTK_LAUNCH_CHECK("loss_laplace_fwd");
}
int ttk_loss_laplace_bwd(const float* mu, const float* b, const float* x, const float* gv, int n, int per, int points, int dim,
                         float chin, float eye, float* gmu, float* gb, ttk_stream_t stream) {
  TTK_REQUIRE(mu && b && x && gv && gmu && gb && n > 0, "loss_laplace_bwd: bad arguments");
  const int total = n * (points ? 204 : per);
  hipLaunchKernelGGL(loss_laplace_bwd_k, TTK_GRID(total), mu, b, x, gv, n, per, points, dim, chin, eye, gmu, gb);
  TTK_LAUNCH_CHECK("loss_laplace_bwd");
}
int ttk_loss_elem_fwd(const float* p, const float* t, const float* colw, int n, int D, int kind, float beta, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(p && t && colw && v && n > 0 && D > 0 && kind >= 0 && kind <= 2 && (kind != 2 || beta > 0.f), "loss_elem_fwd: bad arguments");
  hipLaunchKernelGGL(loss_elem_fwd_k, TTK_GRID(n * 64), p, t, colw, n, D, kind, beta, v);
  TTK_LAUNCH_CHECK("loss_elem_fwd");
}
int ttk_loss_elem_bwd(const float* p, const float* t, const float* colw, const float* gv, int n, int D, int kind, float beta, float* gp,
                      ttk_stream_t stream) {
  TTK_REQUIRE(p && t && colw && gv && gp && n > 0 && D > 0 && kind >= 0 && kind <= 2 && (kind != 2 || beta > 0.f), "loss_elem_bwd: bad arguments");
  hipLaunchKernelGGL(loss_elem_bwd_k, TTK_GRID(n * D), p, t, colw, gv, n, D, kind, beta, gp);
  TTK_LAUNCH_CHECK("loss_elem_bwd");
}
int ttk_loss_rot_geodesic_fwd(const float* q, const float* t, int n, float* v, ttk_stream_t stream) {
  TTK_REQUIRE(q && t && v && n > 0, "loss_rot_geodesic_fwd: bad arguments");
  hipLaunchKernelGGL(loss_rot_geodesic_fwd_k, TTK_GRID(n), q, t, n, v);
  TTK_LAUNCH_CHECK("loss_rot_geodesic_fwd");
}
int ttk_loss_rot_geodesic_bwd(const float* q, const float* t, const float* gv, int n, float* gq, ttk_stream_t stream) {
  TTK_REQUIRE(q && t && gv && gq && n > 0, "loss_rot_geodesic_bwd: bad arguments");
  hipLaunchKernelGGL(loss_rot_geodesic_bwd_k, TTK_GRID(n), q, t, gv, n, gq);
  TTK_LAUNCH_CHECK("loss_rot_geodesic_bwd");
}
int ttk_loss_gmm_fwd(const float* x, const double* ck, const double* mu, const double* sinv, int K, double fudge, int n,
                     float* v, double* post, ttk_stream_t stream) {
  TTK_REQUIRE(x && ck && mu && sinv && v && post && n > 0 && K > 0 && K <= 16, "loss_gmm_fwd: bad arguments");
  hipLaunchKernelGGL(loss_gmm_fwd_k, TTK_GRID(n * 64), x, ck, mu, sinv, K, fudge, n, v, post);
  TTK_LAUNCH_CHECK("loss_gmm_fwd");
}
int ttk_loss_gmm_bwd(const float* x, const double* mu, const double* sinv, const double* post, int K, double fudge,
                     const float* gv, int n, float* gx, ttk_stream_t stream) {
  TTK_REQUIRE(x && mu && sinv && post && gv && gx && n > 0 && K > 0, "loss_gmm_bwd: bad arguments");
  hipLaunchKernelGGL(loss_gmm_bwd_k, TTK_GRID(n * 50), x, mu, sinv, post, K, fudge, gv, n, gx);
  TTK_LAUNCH_CHECK("loss_gmm_bwd");
}

int ttk_loss_batch(int nops, const ttk_loss_op* ops, ttk_stream_t stream) {
  TTK_REQUIRE(nops > 0 && nops <= kBatchMax && ops, "loss_batch: 1..%d ops", kBatchMax);
  BatchArgs a{};
  int items = 0;
  for (int k = 0; k < nops; ++k) {
    TTK_REQUIRE(ops[k].kind >= 0 && ops[k].kind < TTK_OP_COUNT && ops[k].items > 0, "loss_batch: op %d: bad kind or item count", k);
    a.op[k] = ops[k];
    if (ops[k].items > items) items = ops[k].items;
  }
  hipLaunchKernelGGL(loss_batch_k, dim3((unsigned)((items + 255) / 256), (unsigned)nops), dim3(256), 0, (hipStream_t)stream, a);
  TTK_LAUNCH_CHECK("loss_batch");
}

}  // extern "C"
