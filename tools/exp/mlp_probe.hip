// Memory-level-parallelism probe (not part of the product): how many bytes per clock can ONE CU stream from HBM / L2 as a
// function of waves per CU and independent 16-byte loads in flight per lane?  One workgroup per CU (grid = 256 * wgs).
// Build: hipcc --offload-arch=gfx950 -O3 -o mlp_probe mlp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int U>
__global__ void __launch_bounds__(1024) probe(const f32x4* __restrict__ src, float* __restrict__ sink, size_t n_vec, int iters) {
  // every block streams its own contiguous slab; lanes read consecutive 16-byte pieces (whole 1 KB per wave-load)
  const size_t per_block = n_vec / gridDim.x;
  const f32x4* p = src + per_block * blockIdx.x + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = blockDim.x;
  const size_t steps = per_block / (stride * U);
  for (int it = 0; it < iters; ++it) {
    const f32x4* q = p;
    for (size_t s = 0; s < steps; ++s) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(q + u * stride);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
      q += stride * U;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int U>
static void run(const f32x4* d, float* sink, size_t bytes, int threads, int wg_per_cu, const char* what) {
  const int grid = 256 * wg_per_cu;
  const size_t n_vec = bytes / 16;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const int iters = bytes < (size_t)64 << 20 ? 200 : 4;
  hipLaunchKernelGGL(probe<U>, dim3(grid), dim3(threads), 0, 0, d, sink, n_vec, 1);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(probe<U>, dim3(grid), dim3(threads), 0, 0, d, sink, n_vec, iters);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const size_t per_block = n_vec / grid, steps = per_block / ((size_t)threads * U);
  const double moved = (double)steps * threads * U * 16.0 * grid * iters;
  printf("%-4s waves/CU %2d  loads in flight/lane %2d : %7.2f TB/s  (%5.1f B/clk/CU at 2.4 GHz)\n", what, threads / 64 * wg_per_cu, U,
         moved / ms / 1e9, moved / ms / 1e6 / 256 / 2.4e3);
}

int main() {
  const size_t big = (size_t)4 << 30, small = (size_t)24 << 20;  // HBM-resident / L2+MALL-resident
  f32x4* d; float* sink;
  CHECK(hipMalloc(&d, big)); CHECK(hipMalloc(&sink, 16));
  CHECK(hipMemset(d, 0, big));
  for (int pass = 0; pass < 2; ++pass) {
    const size_t bytes = pass ? small : big;
    const char* what = pass ? "L2" : "HBM";
    for (int threads : {256, 512, 1024}) {
      run<1>(d, sink, bytes, threads, 1, what); run<2>(d, sink, bytes, threads, 1, what); run<4>(d, sink, bytes, threads, 1, what);
      run<8>(d, sink, bytes, threads, 1, what); run<16>(d, sink, bytes, threads, 1, what);
    }
    run<4>(d, sink, bytes, 1024, 2, what); run<8>(d, sink, bytes, 1024, 2, what);
  }
  return 0;
}
