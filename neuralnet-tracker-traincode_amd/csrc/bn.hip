// BatchNorm statistic finalisation (tiny kernels; fp64 accumulation over the workgroup partials).
// Reference: nn.BatchNorm2d in training mode, backbones/mobilenet_v1.py:30,66,68,79,84,125,162.
#include <stdarg.h>
#include <string.h>

#include "bc_common.h"  // (fold_rows_fast_body: the combined finalisation + fold launch of the bf16-compute path)

namespace ttk {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// block = 32 channels x 32 row-lanes (1024 threads); each thread strides over the partial rows, 4 loads in flight.
// The partial sums are those of y - pivot[c] (pivot == nullptr: of y): mean = pivot + S1/n, var = S2/n - (S1/n)^2.  With the running
// mean as the pivot the cancellation in the variance is that of a batch whose mean is (running_mean_of_the_batches - pivot), not of
// one whose mean is |mean| / sigma standard deviations from zero.  pivot may alias rmean: the owning thread reads it before it
// writes the update.
__global__ void __launch_bounds__(1024) bn_fwd_finalize_k(const float* __restrict__ part, const float* pivot, int rows, int C, double inv_count,
                                                          double unbias, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, int64_t* nbt, float momentum, float eps,
                                                          float* __restrict__ bn) {
  __shared__ double sm[2][32][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int r = rl; r < rows; r += 32) {
      a += (double)part[((size_t)r * 2 + 0) * C + c];
      b += (double)part[((size_t)r * 2 + 1) * C + c];
    }
  }
  sm[0][rl][cl] = a;
  sm[1][rl][cl] = b;
  __syncthreads();
  float bound = 0.f;
  if (rl == 0 && c < C) {
    for (int i = 1; i < 32; ++i) { a += sm[0][i][cl]; b += sm[1][i][cl]; }
    const double shifted = a * inv_count;
    const double mean = shifted + (pivot ? (double)pivot[c] : 0.0);
    double var = b * inv_count - shifted * shifted;  // biased variance, used for normalisation
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * rstd;
    bn[TTK_BN_SCALE * C + c] = (float)sc;
    bn[TTK_BN_BETA * C + c] = beta[c];
    bn[TTK_BN_MEAN * C + c] = (float)mean;
    bn[TTK_BN_RSTD * C + c] = (float)rstd;
    if (rmean) {  // (after the read of pivot[c] above)
      rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
      rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * var * unbias);
    }
    // bound of |max(scale*(y-mean)+beta, 0)| over the batch (Cauchy-Schwarz: |y-mean| <= sqrt(sum (y-mean)^2)); the
    // variance comes from fp32 partial sums through E[y^2]-mean^2, so it is padded by its possible cancellation error
    const double dev = sqrt((var + 1.0e-6 * (b * inv_count)) / inv_count);
    bound = (float)(fabs(sc) * dev + fabs((double)beta[c])) * 1.0001f;
  }
  if (rl == 0) {  // lanes 0..31 of wave 0
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) bound = fmaxf(bound, __shfl_xor(bound, off));
    if (cl == 0) atomicMax(reinterpret_cast<unsigned*>(bn + (size_t)TTK_BN_AUX * C + TTK_AUX_ACT_BOUND), __float_as_uint(bound));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
}

__global__ void bn_eval_prepare_k(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                                  int C, float* bn) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.0f / sqrtf(rvar[c] + eps);
  bn[TTK_BN_SCALE * C + c] = gamma[c] * rstd;
  bn[TTK_BN_BETA * C + c] = beta[c];
  bn[TTK_BN_MEAN * C + c] = rmean[c];
  bn[TTK_BN_RSTD * C + c] = rstd;
}

// dy = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)),  xhat = (y - mean)*rstd
//    = ga*(g - gmean) + gb*(y - mean)          ga = gamma*rstd, gb = -ga*rstd^2*mean(g*(y-mean))
// partials: sum(g), sum(g*(y - mean));  dgamma = rstd*sum(g*(y-mean)), dbeta = sum(g)
__device__ __forceinline__ void bn_bwd_finalize_body(const float* __restrict__ part, int rows, int C, double inv_count,
                                                     const float* __restrict__ gamma, float* __restrict__ bn,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int accumulate, unsigned block) {
  __shared__ double sm[2][32][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = block * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int r = rl; r < rows; r += 32) {
      a += (double)part[((size_t)r * 2 + 0) * C + c];
      b += (double)part[((size_t)r * 2 + 1) * C + c];
    }
  }
  sm[0][rl][cl] = a;
  sm[1][rl][cl] = b;
  __syncthreads();
  float bound = 0.f;
  if (rl == 0 && c < C) {
    for (int i = 1; i < 32; ++i) { a += sm[0][i][cl]; b += sm[1][i][cl]; }
    const double rs = bn[TTK_BN_RSTD * C + c], ga = gamma[c];
    const double sum_g = a, sum_gx = rs * b;
    const double A = ga * rs;
    bn[TTK_BN_GA * C + c] = (float)A;
    bn[TTK_BN_GB * C + c] = (float)(-A * rs * rs * b * inv_count);
    bn[TTK_BN_GMEAN * C + c] = (float)(sum_g * inv_count);
    if (dgamma) {
      if (accumulate) { dgamma[c] += (float)sum_gx; dbeta[c] += (float)sum_g; }
      else            { dgamma[c] = (float)sum_gx;  dbeta[c] = (float)sum_g; }
    }
    // bound of |ga*(g-gmean) + gb*(y-mean)|: max|g| from the producer of g, |y-mean| <= sqrt(count*var) <= sqrt(count)/rstd
    const double gmax = (double)bn[(size_t)TTK_BN_AUX * C + TTK_AUX_GMAX];
    if (gmax > 0.0)
      bound = (float)(fabs(A) * (gmax + fabs(sum_g * inv_count)) + fabs(A * rs * rs * b * inv_count) * sqrt(1.0 / inv_count) / rs) * 1.0001f;
  }
  if (rl == 0) {
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) bound = fmaxf(bound, __shfl_xor(bound, off));
    if (cl == 0 && bound > 0.f) atomicMax(reinterpret_cast<unsigned*>(bn + (size_t)TTK_BN_AUX * C + TTK_AUX_DY_BOUND), __float_as_uint(bound));
  }
}
__global__ void __launch_bounds__(1024) bn_bwd_finalize_k(const float* __restrict__ part, int rows, int C, double inv_count,
                                                          const float* __restrict__ gamma, float* __restrict__ bn,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          int accumulate) {
  bn_bwd_finalize_body(part, rows, C, inv_count, gamma, bn, dgamma, dbeta, accumulate, blockIdx.x);
}
// The same finalisation AND a fold of workgroup rows (the fused depthwise weight gradient of the kernel that produced `part`) in ONE launch: the first
// nfin blocks finalise, the others fold.  The two are independent and each is a few microseconds of dependent latency: as two launches they cost two
// launch-to-launch round trips of the backward chain, 13 times per step (bf16-compute path).
__global__ void __launch_bounds__(1024) bn_bwd_finalize_fold_k(const float* __restrict__ part, int rows, int C, double inv_count,
                                                               const float* __restrict__ gamma, float* __restrict__ bn,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, unsigned nfin,
                                                               const float* __restrict__ fold_partial, int fold_rows, int64_t fold_n,
                                                               float* __restrict__ fold_out, int fold_accumulate) {
  if (blockIdx.x < nfin) bn_bwd_finalize_body(part, rows, C, inv_count, gamma, bn, dgamma, dbeta, accumulate, blockIdx.x);
  else if (fold_accumulate >= 0) bc::fold_rows_fast_body(fold_partial, fold_rows, fold_n, fold_out, fold_accumulate, blockIdx.x - nfin);
  else bc::fold_rows_wide_body(fold_partial, fold_rows, fold_n, fold_out, blockIdx.x - nfin, 1024);  // (many outputs: always accumulates)
}

// Backward through a FROZEN BatchNorm (eval-mode statistics, reference modelcomponents.py:208-215 freeze_norm_stats: the layer
// is the fixed affine map scale*(y - mean) + beta): dy = scale * g, so ga = scale, gb = 0, gmean = 0; the bound of |dy| is
// max|scale| * max|g|.  One workgroup.
__global__ void __launch_bounds__(256) bn_bwd_frozen_k(float* __restrict__ bn, int C) {
  __shared__ float sm[4];
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float sc = bn[TTK_BN_SCALE * C + c];
    bn[TTK_BN_GA * C + c] = sc;
    bn[TTK_BN_GB * C + c] = 0.f;
    bn[TTK_BN_GMEAN * C + c] = 0.f;
    m = fmaxf(m, fabsf(sc));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float gmax = bn[(size_t)TTK_BN_AUX * C + TTK_AUX_GMAX];
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (gmax > 0.f) bn[(size_t)TTK_BN_AUX * C + TTK_AUX_DY_BOUND] = m * gmax * 1.0001f;
  }
}

// Forward half of the frozen mode: the constants come from the running statistics (bn_eval_prepare_k), but the fp16 GEMMs still
// want a bound of relu(scale*(y - mean_run) + beta) over THIS batch.  From the batch sums (mean_b, var_b):
//   |scale*(y - mean_run) + beta| <= |scale| * (sqrt(count * var_b) + |mean_b - mean_run|) + |beta|     (Cauchy-Schwarz, as above)
__global__ void __launch_bounds__(1024) bn_frozen_bound_k(const float* __restrict__ part, const float* __restrict__ pivot, int rows, int C, double inv_count,
                                                          float* __restrict__ bn) {
  __shared__ double sm[2][32][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int r = rl; r < rows; r += 32) {
      a += (double)part[((size_t)r * 2 + 0) * C + c];
      b += (double)part[((size_t)r * 2 + 1) * C + c];
    }
  }
  sm[0][rl][cl] = a;
  sm[1][rl][cl] = b;
  __syncthreads();
  float bound = 0.f;
  if (rl == 0 && c < C) {
    for (int i = 1; i < 32; ++i) { a += sm[0][i][cl]; b += sm[1][i][cl]; }
    const double shifted = a * inv_count;
    const double mean = shifted + (pivot ? (double)pivot[c] : 0.0);
    double var = b * inv_count - shifted * shifted;
    if (var < 0.0) var = 0.0;
    const double dev = sqrt((var + 1.0e-6 * (b * inv_count)) / inv_count);
    bound = (float)(fabs((double)bn[TTK_BN_SCALE * C + c]) * (dev + fabs(mean - (double)bn[TTK_BN_MEAN * C + c])) + fabs((double)bn[TTK_BN_BETA * C + c])) * 1.0001f;
  }
  if (rl == 0) {
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) bound = fmaxf(bound, __shfl_xor(bound, off));
    if (cl == 0) atomicMax(reinterpret_cast<unsigned*>(bn + (size_t)TTK_BN_AUX * C + TTK_AUX_ACT_BOUND), __float_as_uint(bound));
  }
}

// rows r, r+F, r+2F, ... are summed into row r (r < F), in place: thread (r, i) owns element i of all
// rows congruent to r, so no other thread reads or writes what it touches.
constexpr int kFoldRows = 1024;
__global__ void __launch_bounds__(256) bn_fold_rows_k(float* __restrict__ part, int rows, int row_floats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (i >= row_floats) return;
  float acc = part[(size_t)r * row_floats + i];
  float comp = 0.f;  // Kahan: the folded rows feed an fp64 reduction, keep them clean
  for (int q = r + kFoldRows; q < rows; q += kFoldRows) {
    const float v = part[(size_t)q * row_floats + i] - comp;
    const float t = acc + v;
    comp = (t - acc) - v;
    acc = t;
  }
  part[(size_t)r * row_floats + i] = acc;
}

static int fold_if_needed(float* part, int rows, int C, hipStream_t st) {
  if (rows <= kFoldRows + kFoldRows / 4) return rows;  // e.g. the 1 156 rows of the 17 x 17 layers at B = 512: the finalisation reads them as they are
  hipLaunchKernelGGL(bn_fold_rows_k, dim3((2 * C + 255) / 256, kFoldRows), dim3(256), 0, st, part, rows, 2 * C);
  return kFoldRows;
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_abi_version(void) { return TTK_ABI_VERSION; }
/* hipGetLastError() is per thread and STICKY: an error left behind by any earlier HIP call of the process (device probing
 * during start-up, another library) would be reported by the next entry point's launch check.  The host clears it before a
 * call; returns what was pending. */
int ttk_clear_error(void) { return (int)hipGetLastError(); }
const char* ttk_last_error_string(void) { return ttk::g_err; }

int ttk_partial_rows_elementwise(int64_t work_items) { return elementwise_grid(work_items); }
int ttk_partial_rows_gemm(int64_t M) { return (int)ceil_div(M, TTK_GEMM_BLOCK_M); }

int ttk_bn_fwd_finalize(float* part, const float* pivot, int part_rows, int C, int64_t count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                        float* bn, ttk_stream_t stream) {
  TTK_REQUIRE(part && gamma && beta && bn, "bn_fwd_finalize: null pointer");
  TTK_REQUIRE(C > 0 && part_rows > 0 && count > 0, "bn_fwd_finalize: bad sizes C=%d rows=%d count=%lld", C, part_rows, (long long)count);
  TTK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_fwd_finalize: running_mean/var must both be given");
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  part_rows = fold_if_needed(part, part_rows, C, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_fwd_finalize_k, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, pivot, part_rows, C,
                     1.0 / (double)count, unbias, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                     bn);
  TTK_LAUNCH_CHECK("bn_fwd_finalize");
}

int ttk_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                        int C, float* bn, ttk_stream_t stream) {
  TTK_REQUIRE(gamma && beta && running_mean && running_var && bn, "bn_eval_prepare: null pointer");
  TTK_REQUIRE(C > 0, "bn_eval_prepare: C=%d", C);
  hipLaunchKernelGGL(bn_eval_prepare_k, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, eps, C, bn);
  TTK_LAUNCH_CHECK("bn_eval_prepare");
}

int ttk_bn_bwd_finalize(float* part, int part_rows, int C, int64_t count, const float* gamma, float* bn, float* dgamma,
                        float* dbeta, int accumulate, ttk_stream_t stream) {
  TTK_REQUIRE(part && gamma && bn, "bn_bwd_finalize: null pointer");
  TTK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_bwd_finalize: dgamma/dbeta must both be given");
  TTK_REQUIRE(C > 0 && part_rows > 0 && count > 0, "bn_bwd_finalize: bad sizes");
  part_rows = fold_if_needed(part, part_rows, C, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, part_rows, C,
                     1.0 / (double)count, gamma, bn, dgamma, dbeta, accumulate);
  TTK_LAUNCH_CHECK("bn_bwd_finalize");
}

int ttk_bc_bn_bwd_finalize_fold(float* part, int part_rows, int C, int64_t count, const float* gamma, float* bn, float* dgamma, float* dbeta,
                                int accumulate, const float* fold_partial, int fold_rows, int64_t fold_n, float* fold_out, int fold_accumulate,
                                ttk_stream_t stream) {
  TTK_REQUIRE(part && gamma && bn && fold_partial && fold_out, "bc_bn_bwd_finalize_fold: null pointer");
  TTK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bc_bn_bwd_finalize_fold: dgamma/dbeta must both be given");
  TTK_REQUIRE(C > 0 && part_rows > 0 && count > 0 && fold_rows > 0 && fold_n > 0, "bc_bn_bwd_finalize_fold: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  part_rows = fold_if_needed(part, part_rows, C, st);
  const unsigned nfin = (unsigned)((C + 31) / 32);
  if (bc::fold_rows_fast_ok(fold_rows, fold_n)) {  // few outputs, many rows (depthwise workgroup rows, slice tiles of the early pointwise layers)
    hipLaunchKernelGGL(bn_bwd_finalize_fold_k, dim3(nfin + bc::fold_rows_fast_blocks(fold_n)), dim3(1024), 0, st, part, part_rows, C, 1.0 / (double)count, gamma, bn,
                       dgamma, dbeta, accumulate, nfin, fold_partial, fold_rows, fold_n, fold_out, fold_accumulate);
  } else if (fold_n % 4 == 0 && fold_accumulate) {  // many outputs (slice tiles of the wide pointwise layers)
    hipLaunchKernelGGL(bn_bwd_finalize_fold_k, dim3(nfin + (unsigned)ceil_div(fold_n, 4096)), dim3(1024), 0, st, part, part_rows, C, 1.0 / (double)count, gamma, bn,
                       dgamma, dbeta, accumulate, nfin, fold_partial, fold_rows, fold_n, fold_out, -1);
  } else {  // shapes neither form takes: two launches, same results
    launch_fold_partials(fold_partial, fold_rows, fold_n, fold_out, fold_accumulate, st);
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(nfin), dim3(1024), 0, st, part, part_rows, C, 1.0 / (double)count, gamma, bn, dgamma, dbeta, accumulate);
  }
  TTK_LAUNCH_CHECK("bc_bn_bwd_finalize_fold");
}

int ttk_bn_frozen_bound(float* part, const float* pivot, int part_rows, int C, int64_t count, float* bn, ttk_stream_t stream) {
  TTK_REQUIRE(part && bn && C > 0 && part_rows > 0 && count > 0, "bn_frozen_bound: bad arguments");
  part_rows = fold_if_needed(part, part_rows, C, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_frozen_bound_k, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, pivot, part_rows, C, 1.0 / (double)count, bn);
  TTK_LAUNCH_CHECK("bn_frozen_bound");
}

int ttk_bn_bwd_frozen(float* bn, int C, ttk_stream_t stream) {
  TTK_REQUIRE(bn && C > 0, "bn_bwd_frozen: bad arguments");
  hipLaunchKernelGGL(bn_bwd_frozen_k, dim3(1), dim3(256), 0, (hipStream_t)stream, bn, C);
  TTK_LAUNCH_CHECK("bn_bwd_frozen");
}

}  // extern "C"
