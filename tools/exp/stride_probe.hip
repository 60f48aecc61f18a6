// Row-stride probe (not part of the product): the GEMM producers read, per k32 step, a 128-byte piece of each of 128
// rows that are K*4 bytes apart.  Does a power-of-two row stride (2 KB for K = 512) camp on a subset of the L2/HBM
// channels?  One 256-thread workgroup per CU streams tiles of 128 rows x K floats, k32 step by k32 step, 4 (or 8)
// 16-byte loads per lane in flight, with the row stride K*4 bytes or K*4 + PAD bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -o stride_probe stride_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int U>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ src, float* sink, long rows, int K, long stride_f, int wg_mult) {
  const int tid = threadIdx.x, r0 = tid >> 3, q = tid & 7;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long tiles = rows / 128;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const float* base = src + (t * 128 + r0) * stride_f + q * 4;
    for (int ks = 0; ks < K / 32; ++ks) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (long)(32 * u % 128) * stride_f + ks * 32 + (u / 4) * 0);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int U>
static void run(const float* d, float* sink, long rows, int K, int pad_f, int wg_per_cu) {
  const long stride_f = K + pad_f;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<U>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, sink, rows, K, stride_f, wg_per_cu);
  CHECK(hipEventRecord(a));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<U>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, sink, rows, K, stride_f, wg_per_cu);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double moved = (double)(rows / 128) * 128 * K * 4 * 3;
  printf("K=%4d row stride %5ld B  loads in flight %d  WG/CU %d : %6.2f TB/s  (%5.1f B/clk/CU at 2.1 GHz)\n", K, stride_f * 4, U, wg_per_cu,
         moved / ms / 1e9, moved / ms / 1e6 / 256 / 2.1e3);
}

int main() {
  const size_t bytes = (size_t)3 << 30;
  float *d, *sink;
  CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&sink, 16));
  CHECK(hipMemset(d, 0, bytes));
  for (int K : {512, 1024, 256}) {
    for (int pad : {0, 32, 64}) {
      const long rows = (long)(((size_t)2 << 30) / ((K + pad) * 4)) / 128 * 128;
      run<4>(d, sink, rows, K, pad, 1);
      run<4>(d, sink, rows, K, pad, 2);
    }
  }
  return 0;
}
