// Sustained rate of the two fp16 MFMA shapes under the board's power limit (no memory traffic: registers only).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_power.hip -o tools/exp/_build/mfma_power && tools/exp/_build/mfma_power <0|1> <seconds> [waves per SIMD: 1|2]
// 0: v_mfma_f32_16x16x32_f16 (what the fp32 path's split GEMMs issue), 1: v_mfma_f32_32x32x16_f16.  Prints TFLOP/s per 0.5 s window; sample rocm-smi beside it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
  h8 a[8], b;  // (a distinct A operand per accumulator: with one, hipcc sees eight identical products and chains their registers)
  for (int i = 0; i < 8; ++i) {
    b[i] = (_Float16)(0.002f * (threadIdx.x - i));
    for (int j = 0; j < 8; ++j) a[j][i] = (_Float16)(0.001f * (threadIdx.x + i + 3 * j));
  }
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a[j]), "v"(b));  // (asm: in-place accumulators; hipcc rotates the builtin's registers into a dependency chain here)
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];  // (every element: else hipcc overlaps the accumulators' registers)
  } else {
    f16v c[4];
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) c[j][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a[j]), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) s += c[j][k];
  }
  if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 0;
  const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
  const int wps = argc > 3 ? atoi(argv[3]) : 2;  // waves per SIMD
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000, blocks = 256 * wps;  // 256 CUs x wps workgroups of 4 waves
  const double flop_per_launch = (double)blocks * 4 * iters * (shape == 0 ? 8 * 16384.0 : 4 * 32768.0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) {
      if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("shape %d waves/SIMD %d: %.0f TFLOP/s (%.2f ms per 20 launches)\n", shape, wps, 20 * flop_per_launch / (ms * 1e-3) / 1e12, ms);
    fflush(stdout);
  }
  return 0;
}
