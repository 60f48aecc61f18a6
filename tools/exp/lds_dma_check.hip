#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* __restrict__ g, u32x4* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane reads a permuted global element; lands at lds + wave*1024 + lane*16
  const u32x4* src = g + wave * 64 + (lane ^ 1);
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                   (void __attribute__((address_space(3)))*)(lds + wave * 1024), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[threadIdx.x] = *reinterpret_cast<u32x4*>(lds + threadIdx.x * 16);
}
int main() {
  u32x4 *g, *o;
  hipMalloc(&g, 256 * 16); hipMalloc(&o, 256 * 16);
  unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o);
  unsigned r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) for (int j = 0; j < 4; ++j) if (r[t * 4 + j] != (unsigned)(((t & ~63) + ((t & 63) ^ 1)) * 4 + j)) ++bad;
  printf("bad=%d r[0..7]= %u %u %u %u %u %u %u %u\n", bad, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
  return bad != 0;
}
