// Access-pattern probe (not part of the product): HBM throughput of the depthwise kernels' tile pattern - every 8 lanes
// read/write one 128-byte segment (32 channels of one pixel), consecutive pixels C*4 bytes apart, the C/32 slabs of a
// pixel handled by neighbouring workgroups - against fully contiguous streaming.  Read-only and read+write (copy).
// Build: hipcc --offload-arch=gfx950 -O3 -o slab_probe slab_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// SEGQ = 16-byte quads per contiguous segment (8 = 128 B slab of 32 channels, 16 = 256 B, 64 = 1 KB)
template <int U, int SEGQ, bool COPY, bool NT = false>
__global__ void __launch_bounds__(256) probe(const f32x4* __restrict__ src, f32x4* __restrict__ dst, float* sink, size_t npix, int C) {
  const int cq = C / 4;                       // quads per pixel
  const int nslabs = cq / SEGQ;
  const int slab = blockIdx.x % nslabs;
  const size_t group = blockIdx.x / nslabs, ngroups = gridDim.x / nslabs;
  const int q = threadIdx.x % SEGQ, slot = threadIdx.x / SEGQ, nslot = 256 / SEGQ;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // pixels are dealt to groups in chunks of 1024 (a "band"), like the tiles of the real kernels
  for (size_t base = group * 1024; base < npix; base += ngroups * 1024) {
    for (int p = slot; p < 1024; p += nslot * U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[(base + p + u * nslot) * cq + slab * SEGQ + q]) : src[(base + p + u * nslot) * cq + slab * SEGQ + q];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (COPY) { if (NT) __builtin_nontemporal_store(v[u] * 2.f, &dst[(base + p + u * nslot) * cq + slab * SEGQ + q]); else dst[(base + p + u * nslot) * cq + slab * SEGQ + q] = v[u] * 2.f; }
        else acc += v[u];
      }
    }
  }
  if (!COPY && acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int U, int SEGQ, bool COPY, bool NT = false>
static void run(const f32x4* s, f32x4* d, float* sink, size_t npix, int C, int wg_per_cu) {
  const int nslabs = C / 4 / SEGQ;
  int grid = 256 * wg_per_cu / nslabs * nslabs;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((probe<U, SEGQ, COPY, NT>), dim3(grid), dim3(256), 0, 0, s, d, sink, npix, C);
  CHECK(hipEventRecord(a));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<U, SEGQ, COPY, NT>), dim3(grid), dim3(256), 0, 0, s, d, sink, npix, C);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double moved = (double)npix * C * 4 * (COPY ? 2 : 1) * 3;
  printf("%s%s C=%4d segment %4d B  loads in flight %d  WG/CU %d : %6.2f TB/s\n", COPY ? "copy" : "read", NT ? "-nt" : "   ", C, SEGQ * 16, U, wg_per_cu, moved / ms / 1e9);
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  f32x4 *s, *d; float* sink;
  CHECK(hipMalloc(&s, bytes)); CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&sink, 16));
  CHECK(hipMemset(s, 0, bytes));
  for (int C : {512, 128}) {
    const size_t npix = bytes / (C * 4) / 1024 * 1024;
    run<4, 8, false>(s, d, sink, npix, C, 3); run<8, 8, false>(s, d, sink, npix, C, 3); run<4, 8, false>(s, d, sink, npix, C, 8);
    run<4, 16, false>(s, d, sink, npix, C, 3); run<4, 32, false>(s, d, sink, npix, C, 3);
    run<4, 8, true>(s, d, sink, npix, C, 3); run<8, 8, true>(s, d, sink, npix, C, 3); run<4, 8, true>(s, d, sink, npix, C, 8);
    run<4, 16, true>(s, d, sink, npix, C, 3); run<4, 32, true>(s, d, sink, npix, C, 3); run<4, 32, true>(s, d, sink, npix, C, 8);
    run<4, 8, false, true>(s, d, sink, npix, C, 3); run<4, 8, true, true>(s, d, sink, npix, C, 3); run<4, 32, true, true>(s, d, sink, npix, C, 3);
  }
  return 0;
}
