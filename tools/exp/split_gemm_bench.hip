// Feasibility experiment (not part of the product): fp32 GEMM  Y = act(X) * W^T  on the bf16 MFMA pipe with
// exact 3-way operand splits (x = hi + mid + lo, each piece 8 significant bits) and the 6 products whose
// magnitude is >= 2^-24 of the leading one.  Build: hipcc --offload-arch=gfx950 -O3 -o split_gemm_bench split_gemm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                 // bytes per LDS row of one piece plane: 32 bf16 + 16 B pad
constexpr int PLANE = 128 * ROWB;        // 10240
#define PROD(pa, pb)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t pack_hi16(float a, float b) {  // top halves of a (low word) and b (high word)
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// exact 3-way split of 4 floats -> 3 x (2 dwords of packed bf16)
__device__ __forceinline__ void split4(float4 v, uint2& h, uint2& m, uint2& l) {
  float x[4] = {v.x, v.y, v.z, v.w}, r1[4], r2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r1[i] = x[i] - __uint_as_float(__float_as_uint(x[i]) & 0xffff0000u);
    r2[i] = r1[i] - __uint_as_float(__float_as_uint(r1[i]) & 0xffff0000u);
  }
  h = make_uint2(pack_hi16(x[0], x[1]), pack_hi16(x[2], x[3]));
  m = make_uint2(pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3]));
  l = make_uint2(pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3]));
}

__global__ void split_w_k(const float* w, uint16_t* wp, int64_t n) {  // wp[3][n]
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = w[i];
  uint32_t u = __float_as_uint(x);
  float r1 = x - __uint_as_float(u & 0xffff0000u);
  uint32_t u1 = __float_as_uint(r1);
  float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
  wp[i] = u >> 16; wp[n + i] = u1 >> 16; wp[2 * n + i] = __float_as_uint(r2) >> 16;
}

template <int NPROD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_split_k(const float* __restrict__ X, const float* __restrict__ bn,
                                                       const uint16_t* __restrict__ Wp, float* __restrict__ Y,
                                                       int M, int K, int N) {
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* As = lds;               // [3][128][80]
  unsigned char* Bs = lds + 3 * PLANE;   // [3][128][80]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int r = lane & 31, h = lane >> 5;
  const int ntn = N / BN;
  const int tile = blockIdx.x;
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int64_t WN = (int64_t)N * K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[4];
  u32x4 rb[3][2];
  const int arow = t >> 3, akq = t & 7;
#define GLOAD(k0) { \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
      int row = m0 + arow + 32 * i; \
      ra[i] = row < M ? *reinterpret_cast<const float4*>(X + (int64_t)row * K + k0 + akq * 4) : make_float4(0, 0, 0, 0); \
    } \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) \
      _Pragma("unroll") for (int c = 0; c < 2; ++c) { \
        int ch = t + 256 * c, row = ch >> 2, part = ch & 3; \
        rb[p][c] = *reinterpret_cast<const u32x4*>(Wp + p * WN + (int64_t)(n0 + row) * K + k0 + part * 8); \
      } \
  }
#define LSTORE(k0) { \
    const float4 sc = *reinterpret_cast<const float4*>(bn + 0 * K + k0 + akq * 4); \
    const float4 be = *reinterpret_cast<const float4*>(bn + 1 * K + k0 + akq * 4); \
    const float4 mu = *reinterpret_cast<const float4*>(bn + 2 * K + k0 + akq * 4); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
      float4 v = ra[i]; \
      v.x = fmaxf(fmaf(sc.x, v.x - mu.x, be.x), 0.f); v.y = fmaxf(fmaf(sc.y, v.y - mu.y, be.y), 0.f); \
      v.z = fmaxf(fmaf(sc.z, v.z - mu.z, be.z), 0.f); v.w = fmaxf(fmaf(sc.w, v.w - mu.w, be.w), 0.f); \
      uint2 ph, pm, pl; \
      split4(v, ph, pm, pl); \
      int off = (arow + 32 * i) * ROWB + akq * 8; \
      *reinterpret_cast<uint2*>(As + 0 * PLANE + off) = ph; \
      *reinterpret_cast<uint2*>(As + 1 * PLANE + off) = pm; \
      *reinterpret_cast<uint2*>(As + 2 * PLANE + off) = pl; \
    } \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) \
      _Pragma("unroll") for (int c = 0; c < 2; ++c) { \
        int ch = t + 256 * c, row = ch >> 2, part = ch & 3; \
        *reinterpret_cast<u32x4*>(Bs + p * PLANE + row * ROWB + part * 16) = rb[p][c]; \
      } \
  }

  GLOAD(0)
  LSTORE(0)
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = k0 + BK < K;
    const int kn = more ? k0 + BK : k0;
    GLOAD(kn)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = *reinterpret_cast<const bf16x8*>(As + p * PLANE + (wm * 64 + i * 32 + r) * ROWB + kb * 32 + h * 16);
          b[i][p] = *reinterpret_cast<const bf16x8*>(Bs + p * PLANE + (wn * 64 + i * 32 + r) * ROWB + kb * 32 + h * 16);
        }
      // smallest products first: (h,l) (l,h) (m,m) | (h,m) (m,h) | (h,h)
      if constexpr (NPROD >= 6) { PROD(0, 2) PROD(2, 0) PROD(1, 1) }
      if constexpr (NPROD >= 3) { PROD(0, 1) PROD(1, 0) }
      PROD(0, 0)
    }
    __syncthreads();
    if (more) LSTORE(kn)
    __syncthreads();
  }
  // plain epilogue (experiment): lane owns column, 16 rows
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        int col = n0 + wn * 64 + j * 32 + r;
        if (row < M) Y[(int64_t)row * N + col] = acc[i][j][e];
      }
}


// ---- experiment 2: 128x256 tile, 8 waves (2x4, wave tile 64x64), BK=16 stages, LDS double buffer, loads 2 stages ahead.
// Split weights in k-block-major layout Wq[3][K/16][N][16] so a stage of B is one contiguous slab.
constexpr int ROW2 = 48;  // 16 bf16 + 16 B pad: odd multiple of 16 B -> conflict-free ds_read_b128
constexpr int APL2 = 128 * ROW2, BPL2 = 256 * ROW2;
constexpr int STAGE2 = 3 * APL2 + 3 * BPL2;  // 55296

__global__ void split_w2_k(const float* w, uint16_t* wq, int N, int K) {  // w[N][K] -> wq[3][K/16][N][16]
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * K) return;
  int n = i / K, k = i % K;
  float x = w[i];
  uint32_t u = __float_as_uint(x);
  float r1 = x - __uint_as_float(u & 0xffff0000u);
  uint32_t u1 = __float_as_uint(r1);
  float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
  int64_t o = ((int64_t)(k >> 4) * N + n) * 16 + (k & 15), pl = (int64_t)N * K;
  wq[o] = u >> 16; wq[pl + o] = u1 >> 16; wq[2 * pl + o] = __float_as_uint(r2) >> 16;
}

template <int NPROD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_split2_k(const float* __restrict__ X, const float* __restrict__ bn, const uint16_t* __restrict__ Wq,
                   float* __restrict__ Y, int M, int K, int N, int flags) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int r = lane & 31, h = lane >> 5;
  const unsigned G = gridDim.x, Lid = blockIdx.x, NB = N / 256;
  const unsigned xq = G / 8, xr = G % 8, xcd = Lid % 8;
  const unsigned tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const int m0 = (tile / NB) * 128, n0 = (tile % NB) * 256;
  const int64_t WPL = (int64_t)N * K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // A: 128 rows x 16 k fp32 = 512 float4 -> one per thread.  B: per piece 256 rows x 32 B = 512 x 16 B -> one per thread.
  const int arow = t >> 2, akq = t & 3;
  const int brow = t >> 1, bhalf = t & 1;
  const float* aptr = X + (int64_t)min(((flags & 2) ? 0 : m0) + arow, M - 1) * K + akq * 4;
  const uint16_t* bptr = Wq + ((int64_t)n0 + brow) * 16 + bhalf * 8;
  const int a_lds = arow * ROW2 + akq * 8;
  const int b_lds = 3 * APL2 + brow * ROW2 + bhalf * 16;
  const int a_rd = (wm * 64 + r) * ROW2 + h * 16;
  const int b_rd = 3 * APL2 + (wn * 64 + r) * ROW2 + h * 16;

  float4 ra[2];
  u32x4 rb[2][3];
#define GLOAD2(set, k0) {                                                                                   \
    ra[set] = *reinterpret_cast<const float4*>(aptr + (k0));                                                \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                           \
      rb[set][p] = *reinterpret_cast<const u32x4*>(bptr + p * WPL + (int64_t)((flags & 4) ? 0 : ((k0) >> 4)) * N * 16);          \
  }
#define LSTORE2(set, k0, buf) {                                                                             \
    unsigned char* S = lds + (buf) * STAGE2;                                                                \
    const float4 sc = *reinterpret_cast<const float4*>(bnS + 0 * K + (k0) + akq * 4);                       \
    const float4 be = *reinterpret_cast<const float4*>(bnS + 1 * K + (k0) + akq * 4);                       \
    const float4 mu = *reinterpret_cast<const float4*>(bnS + 2 * K + (k0) + akq * 4);                       \
    float4 v = ra[set];                                                                                     \
    v.x = fmaxf(fmaf(sc.x, v.x - mu.x, be.x), 0.f); v.y = fmaxf(fmaf(sc.y, v.y - mu.y, be.y), 0.f);         \
    v.z = fmaxf(fmaf(sc.z, v.z - mu.z, be.z), 0.f); v.w = fmaxf(fmaf(sc.w, v.w - mu.w, be.w), 0.f);         \
    uint2 ph, pm, pl;                                                                                       \
    split4(v, ph, pm, pl);                                                                                  \
    *reinterpret_cast<uint2*>(S + 0 * APL2 + a_lds) = ph;                                                   \
    *reinterpret_cast<uint2*>(S + 1 * APL2 + a_lds) = pm;                                                   \
    *reinterpret_cast<uint2*>(S + 2 * APL2 + a_lds) = pl;                                                   \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                           \
      *reinterpret_cast<u32x4*>(S + p * BPL2 + b_lds) = rb[set][p];                                         \
  }
#define COMPUTE2(buf) {                                                                                     \
    const unsigned char* S = lds + (buf) * STAGE2;                                                          \
    bf16x8 a[2][3], b[2][3];                                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int p = 0; p < 3; ++p) {           \
      a[i][p] = *reinterpret_cast<const bf16x8*>(S + p * APL2 + a_rd + i * 32 * ROW2);                      \
      b[i][p] = *reinterpret_cast<const bf16x8*>(S + p * BPL2 + b_rd + i * 32 * ROW2);                      \
    }                                                                                                       \
    if constexpr (NPROD >= 6) { PROD(0, 2) PROD(2, 0) PROD(1, 1) }                                          \
    if constexpr (NPROD >= 3) { PROD(0, 1) PROD(1, 0) }                                                     \
    PROD(0, 0)                                                                                              \
  }
  const int nk = K / 16;
  float* bnS = reinterpret_cast<float*>(lds + 2 * STAGE2);
  for (int i = t; i < 3 * K; i += 512) bnS[i] = bn[i];
  GLOAD2(0, 0)
  GLOAD2(1, 16)
  __syncthreads();
  LSTORE2(0, 0, 0)
  __syncthreads();
  // steady state, unrolled by two so register sets / LDS buffers are compile-time
  for (int s = 0; s < nk; s += 2) {
    // stage s in buf 0 ; set 1 holds stage s+1 ; set 0 is free -> load stage s+2
    { const int kn = min(s + 2, nk - 1) * 16; GLOAD2(0, kn) }
    __builtin_amdgcn_sched_barrier(0);
    COMPUTE2(0)
    LSTORE2(1, (s + 1) * 16, 1)
    __syncthreads();
    // stage s+1 in buf 1 ; set 0 holds stage s+2 ; set 1 free -> load stage s+3
    { const int kn = min(s + 3, nk - 1) * 16; GLOAD2(1, kn) }
    __builtin_amdgcn_sched_barrier(0);
    COMPUTE2(1)
    if (s + 2 < nk) LSTORE2(0, (s + 2) * 16, 0)
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        int col = n0 + wn * 64 + j * 32 + r;
        if (row < M && (!(flags & 1) || acc[i][j][e] == 12345.f)) Y[(int64_t)row * N + col] = acc[i][j][e];
      }
}


// ---- experiment 3: wave-specialised.  512 threads: waves 0-3 consume (ds_read + MFMA, wave tile 64x128 of a 128x256
// block tile), waves 4-7 produce (global fp32 loads of BOTH operands, BN+ReLU on A, exact 3-way bf16 split, LDS writes).
// One consumer + one producer wave per SIMD.  LDS: ring of 2 super-stages (k32 = 2 k16 stages), unpadded 32-B rows,
// chunk swizzle c ^= (row>>3)&1 for conflict-free ds_read_b128.  One barrier per k32.
constexpr int A3 = 128 * 32, B3 = 256 * 32;           // bytes per piece plane of a k16 stage
constexpr int STAGE3 = 3 * A3 + 3 * B3;               // 36864
__device__ __forceinline__ int swz(int row, int c) { return row * 32 + ((c ^ ((row >> 3) & 1)) << 4); }

template <int NPROD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_split3_k(const float* __restrict__ X, const float* __restrict__ bn, const float* __restrict__ W,
                   float* __restrict__ Y, int M, int K, int N, int flags, int lda, int ldb) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int t = threadIdx.x;
  const unsigned G = gridDim.x, Lid = blockIdx.x, NB = N / 256;
  const unsigned xq = G / 8, xr = G % 8, xcd = Lid % 8;
  const unsigned tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const int m0 = (tile / NB) * 128, n0 = (tile % NB) * 256;
  float* bnS = reinterpret_cast<float*>(lds + 4 * STAGE3);
  for (int i = t; i < 3 * K; i += 512) bnS[i] = bn[i];
  const int nks = K / 32;
  const bool producer = __builtin_amdgcn_readfirstlane(t) >= 256;

  if (producer) {
    const int pt = t - 256;
    const int row0 = pt >> 3, kq8 = pt & 7;             // 32 rows per pass, 8 lanes x 16 B per row
    const int sub = kq8 >> 2, kq = kq8 & 3;             // k16 stage within the super-stage, float4 within it
    const int c = kq >> 1, o8 = (kq & 1) * 8;
    float4 ra[4], rb[8];
    for (int i = 0; i < 4; ++i) ra[i] = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
    for (int i = 0; i < 8; ++i) rb[i] = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
    const float* ap = X + (int64_t)m0 * lda + kq8 * 4;
    const float* bp = W + (int64_t)n0 * ldb + kq8 * 4;
#define PLOAD(ks) {                                                                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        int row = row0 + 32 * i;                                                                               \
        if (!(flags & 16)) ra[i] = *reinterpret_cast<const float4*>(ap + (int64_t)min(row, M - 1 - m0) * lda + (ks) * 32);          \
      }                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                            \
        if (!(flags & 16)) rb[i] = *reinterpret_cast<const float4*>(bp + (int64_t)(row0 + 32 * i) * ldb + (ks) * 32);               \
    }
#define PSTORE(ks) {                                                                                           \
      unsigned char* S = lds + (((ks) & 1) * 2 + sub) * STAGE3;                                                \
      const float4 sc = *reinterpret_cast<const float4*>(bnS + 0 * K + (ks) * 32 + kq8 * 4);                   \
      const float4 be = *reinterpret_cast<const float4*>(bnS + 1 * K + (ks) * 32 + kq8 * 4);                   \
      const float4 mu = *reinterpret_cast<const float4*>(bnS + 2 * K + (ks) * 32 + kq8 * 4);                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        float4 v = ra[i];                                                                                      \
        v.x = fmaxf(fmaf(sc.x, v.x - mu.x, be.x), 0.f); v.y = fmaxf(fmaf(sc.y, v.y - mu.y, be.y), 0.f);        \
        v.z = fmaxf(fmaf(sc.z, v.z - mu.z, be.z), 0.f); v.w = fmaxf(fmaf(sc.w, v.w - mu.w, be.w), 0.f);        \
        uint2 ph, pm, pl;                                                                                      \
        split4(v, ph, pm, pl);                                                                                 \
        const int off = swz(row0 + 32 * i, c) + o8;                                                            \
        *reinterpret_cast<uint2*>(S + 0 * A3 + off) = ph;                                                      \
        *reinterpret_cast<uint2*>(S + 1 * A3 + off) = pm;                                                      \
        *reinterpret_cast<uint2*>(S + 2 * A3 + off) = pl;                                                      \
      }                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
        uint2 ph, pm, pl;                                                                                      \
        split4(rb[i], ph, pm, pl);                                                                             \
        const int off = 3 * A3 + swz(row0 + 32 * i, c) + o8;                                                   \
        *reinterpret_cast<uint2*>(S + 0 * B3 + off) = ph;                                                      \
        *reinterpret_cast<uint2*>(S + 1 * B3 + off) = pm;                                                      \
        *reinterpret_cast<uint2*>(S + 2 * B3 + off) = pl;                                                      \
      }                                                                                                        \
    }
    PLOAD(0)
    __syncthreads();  // bnS visible
    PSTORE(0)
    if (nks > 1) PLOAD(1)
    __syncthreads();  // super-stage 0 ready
    for (int it = 0; it < nks; ++it) {
      if (it + 1 < nks && !(flags & 64)) {
        PSTORE(it + 1)
        if (it + 2 < nks) PLOAD(it + 2)
      }
      __syncthreads();
    }
  } else {
    const int lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int a_off[2], b_off[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_off[i] = swz(wm * 64 + i * 32 + r, h);
#pragma unroll
    for (int j = 0; j < 4; ++j) b_off[j] = 3 * A3 + swz(wn * 128 + j * 32 + r, h);
    __syncthreads();
    __syncthreads();
    for (int it = 0; it < nks; ++it) {
      if (!(flags & 32))
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const unsigned char* S = lds + ((it & 1) * 2 + sub) * STAGE3;
        bf16x8 a[2][3], b[4][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i][p] = *reinterpret_cast<const bf16x8*>(S + p * A3 + a_off[i]);
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j][p] = *reinterpret_cast<const bf16x8*>(S + p * B3 + b_off[j]);
        }
#define PROD3(pa, pb)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[i][j] =       \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
        if constexpr (NPROD >= 6) { PROD3(0, 2) PROD3(2, 0) PROD3(1, 1) }
        if constexpr (NPROD >= 3) { PROD3(0, 1) PROD3(1, 0) }
        PROD3(0, 0)
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          int col = n0 + wn * 128 + j * 32 + r;
          if (row < M && (!(flags & 1) || acc[i][j][e] == 12345.f)) Y[(int64_t)row * N + col] = acc[i][j][e];
        }
  }
}

int main(int argc, char** argv) {
  int flags = argc > 4 ? atoi(argv[4]) : 0;
  int M = argc > 1 ? atoi(argv[1]) : 41472, K = argc > 2 ? atoi(argv[2]) : 512, N = argc > 3 ? atoi(argv[3]) : 512;
  const int lda = (flags & 8) ? K + 32 : K, ldb = (flags & 8) ? K + 32 : K;
  std::vector<float> hX((size_t)M * lda), hW((size_t)N * ldb), hbn(3 * K);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : hX) v = rnd();
  for (auto& v : hW) v = rnd() * 0.05f;
  for (int k = 0; k < K; ++k) { hbn[k] = 0.5f + 0.5f * fabsf(rnd()); hbn[K + k] = 0.1f * rnd(); hbn[2 * K + k] = 0.1f * rnd(); }
  float *X, *W, *bn, *Y; uint16_t* Wp;
  CHECK(hipMalloc(&X, hX.size() * 4)); CHECK(hipMalloc(&W, hW.size() * 4)); CHECK(hipMalloc(&bn, hbn.size() * 4));
  CHECK(hipMalloc(&Y, (size_t)M * N * 4)); CHECK(hipMalloc(&Wp, hW.size() * 2 * 3));
  CHECK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(bn, hbn.data(), hbn.size() * 4, hipMemcpyHostToDevice));
  int64_t nw = (int64_t)N * K;
  split_w_k<<<(nw + 255) / 256, 256>>>(W, Wp, nw);
  const int tiles = ((M + BM - 1) / BM) * (N / BN);
  const size_t ldsz = 6 * PLANE;
  CHECK(hipFuncSetAttribute((const void*)gemm_split_k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz));
  CHECK(hipFuncSetAttribute((const void*)gemm_split_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz));
  CHECK(hipFuncSetAttribute((const void*)gemm_split_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz));
  uint16_t* Wq; CHECK(hipMalloc(&Wq, hW.size() * 2 * 3));
  split_w2_k<<<(nw + 255) / 256, 256>>>(W, Wq, N, K);
  const int tiles2 = ((M + 127) / 128) * (N / 256);
  const size_t ldsz2 = 2 * STAGE2 + 3 * K * 4;
  CHECK(hipFuncSetAttribute((const void*)gemm_split2_k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz2));
  CHECK(hipFuncSetAttribute((const void*)gemm_split2_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz2));
  CHECK(hipFuncSetAttribute((const void*)gemm_split2_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz2));
  const size_t ldsz3 = 4 * STAGE3 + 3 * K * 4;
  CHECK(hipFuncSetAttribute((const void*)gemm_split3_k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz3));
  CHECK(hipFuncSetAttribute((const void*)gemm_split3_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz3));
  CHECK(hipFuncSetAttribute((const void*)gemm_split3_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsz3));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> hY((size_t)M * N);
  for (int np : {26, 21}) {
    auto launch = [&] {
      if (np == 26) { gemm_split3_k<6><<<tiles2, 512, ldsz3>>>(X, bn, W, Y, M, K, N, flags, lda, ldb); return; }
      if (np == 23) { gemm_split3_k<3><<<tiles2, 512, ldsz3>>>(X, bn, W, Y, M, K, N, flags, lda, ldb); return; }
      if (np == 21) { gemm_split3_k<1><<<tiles2, 512, ldsz3>>>(X, bn, W, Y, M, K, N, flags, lda, ldb); return; }
      if (np == 16) { gemm_split2_k<6><<<tiles2, 512, ldsz2>>>(X, bn, Wq, Y, M, K, N, flags); return; }
      if (np == 13) { gemm_split2_k<3><<<tiles2, 512, ldsz2>>>(X, bn, Wq, Y, M, K, N, flags); return; }
      if (np == 11) { gemm_split2_k<1><<<tiles2, 512, ldsz2>>>(X, bn, Wq, Y, M, K, N, flags); return; }
      if (np == 6) gemm_split_k<6><<<tiles, 256, ldsz>>>(X, bn, Wp, Y, M, K, N);
      else if (np == 3) gemm_split_k<3><<<tiles, 256, ldsz>>>(X, bn, Wp, Y, M, K, N);
      else gemm_split_k<1><<<tiles, 256, ldsz>>>(X, bn, Wp, Y, M, K, N);
    };
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1000 / 20;
    CHECK(hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost));
    // check 64 sampled rows against double
    double maxerr = 0, maxref = 0, err32 = 0;
    for (int s = 0; s < 64; ++s) {
      int row = (int)(((int64_t)s * 7919 + 13) % M);
      for (int col = 0; col < N; col += 7) {
        double ref = 0; float f32 = 0.f;
        for (int k = 0; k < K; ++k) {
          float a = fmaxf(fmaf(hbn[k], hX[(size_t)row * K + k] - hbn[2 * K + k], hbn[K + k]), 0.f);
          ref += (double)a * hW[(size_t)col * K + k];
          f32 = fmaf(a, hW[(size_t)col * K + k], f32);
        }
        maxerr = fmax(maxerr, fabs(hY[(size_t)row * N + col] - ref));
        err32 = fmax(err32, fabs(f32 - ref));
        maxref = fmax(maxref, fabs(ref));
      }
    }
    printf("products=%d  M=%d K=%d N=%d  %8.1f us  %6.1f TF(fp32-equivalent)  max|err|=%.3e (fp32 fma chain %.3e, max|ref|=%.2f)\n",
           np, M, K, N, us, 2.0 * M * K * N / us / 1e6, maxerr, err32, maxref);
  }
  return 0;
}
