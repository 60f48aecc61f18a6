// Streaming probe: what this box's memory system delivers for the access shapes of the HBM-bound kernels of the step
// (measurement infrastructure behind bench.py's `copy_probe` object and tools/stream_sweep.py; no reference counterpart).
//
// The buffer is seen as rows of `row_bytes`; a workgroup streams SEGMENTS - `seg_bytes` wide column slices - of consecutive
// rows, 16 B per lane, U independent loads per lane and stream in flight before anything is consumed:
//   seg_bytes == row_bytes  : a linear sweep (what a copy kernel does);
//   seg_bytes = 128 | 256   : the depthwise kernels' shape (a 32- / 64-channel slab of a channels-last tensor: one 128- /
//                             256-byte piece per pixel, the pieces row_bytes = 4 C apart).
// nread streams (the same shape, `stream_bytes` apart) are read and summed, nwrite are written: 1/0 = read only,
// 1/1 = copy, 5/1 = the depthwise backward's mix.
#include "ttk_common.h"

namespace ttk {

template <int U, bool NT, int NR>
__global__ void __launch_bounds__(256) stream_probe_k(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ sink,
                                                       int64_t rows, int row_vec, int seg_vec, int nwrite, int64_t stream_elems) {
  const int tid = threadIdx.x;
  const int lane_in_seg = tid % seg_vec, row_in_pass = tid / seg_vec, rows_per_pass = 256 / seg_vec;
  const int slabs = row_vec / seg_vec;
  const int64_t rows_per_unit = (int64_t)rows_per_pass * U;
  const int64_t units = slabs * ceil_div(rows, rows_per_unit);
  float4 acc = f4(0.f);
  for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
    const int slab = (int)(u % slabs);
    const int64_t r0 = (u / slabs) * rows_per_unit + row_in_pass;
    const int64_t col = ((int64_t)slab * seg_vec + lane_in_seg) * 4;
    float4 t[NR][U];  // every load of the unit is issued before the first is consumed: NR * U * 16 B in flight per lane
#pragma unroll
    for (int s = 0; s < NR; ++s)
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int64_t r = r0 + (int64_t)i * rows_per_pass;
        const float* p = src + s * stream_elems + (r < rows ? r : rows - 1) * (int64_t)row_vec * 4 + col;
        t[s][i] = NT ? ld4nt(p) : ld4(p);
      }
    float4 v[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      v[i] = t[0][i];
#pragma unroll
      for (int s = 1; s < NR; ++s) v[i] = add4(v[i], t[s][i]);
    }
    if (nwrite) {
      for (int s = 0; s < nwrite; ++s)
#pragma unroll
        for (int i = 0; i < U; ++i) {
          const int64_t r = r0 + (int64_t)i * rows_per_pass;
          if (r < rows) st4(dst + s * stream_elems + r * (int64_t)row_vec * 4 + col, v[i]);
        }
    } else {
#pragma unroll
      for (int i = 0; i < U; ++i) acc = add4(acc, v[i]);
    }
  }
  if (!nwrite && acc.x + acc.y + acc.z + acc.w == 1.2345678e-30f) sink[0] = acc.x;  // keeps the loads alive; never true on real data
}

}  // namespace ttk

using namespace ttk;

extern "C" int ttk_stream_probe(const float* src, float* dst, float* sink, int64_t rows, int row_bytes, int seg_bytes, int nread, int nwrite,
                                int64_t stream_bytes, int unroll, int nontemporal, int blocks, ttk_stream_t stream) {
  TTK_REQUIRE(src && sink && (dst || !nwrite), "ttk_stream_probe: null pointer");
  TTK_REQUIRE(rows > 0 && seg_bytes >= 16 && seg_bytes <= 4096 && (seg_bytes & (seg_bytes - 1)) == 0 && row_bytes % seg_bytes == 0,
              "ttk_stream_probe: seg_bytes must be a power of two in 16..4096 that divides row_bytes (%d, %d)", seg_bytes, row_bytes);
  TTK_REQUIRE((nread == 1 || nread == 2 || nread == 5) && nwrite >= 0 && stream_bytes % 16 == 0 && blocks >= 1, "ttk_stream_probe: 1, 2 or 5 read streams");
  TTK_REQUIRE(unroll == 1 || unroll == 2 || unroll == 4 || unroll == 8, "ttk_stream_probe: unroll 1, 2, 4 or 8");
  hipStream_t st = (hipStream_t)stream;
#define TTK_PROBE(U_, NT_, NR_)                                                                                                             \
  hipLaunchKernelGGL((stream_probe_k<U_, NT_, NR_>), dim3((unsigned)blocks), dim3(256), 0, st, src, dst, sink, rows, row_bytes / 16, seg_bytes / 16, \
                     nwrite, stream_bytes / 4)
#define TTK_PROBE_U(NT_, NR_)                                                                                                  \
  do {                                                                                                                         \
    if (unroll == 1) TTK_PROBE(1, NT_, NR_); else if (unroll == 2) TTK_PROBE(2, NT_, NR_); else if (unroll == 4) TTK_PROBE(4, NT_, NR_); \
    else TTK_PROBE(8, NT_, NR_);                                                                                               \
  } while (0)
  if (nontemporal) {
    if (nread == 1) TTK_PROBE_U(true, 1); else if (nread == 2) TTK_PROBE_U(true, 2); else TTK_PROBE_U(true, 5);
  } else {
    if (nread == 1) TTK_PROBE_U(false, 1); else if (nread == 2) TTK_PROBE_U(false, 2); else TTK_PROBE_U(false, 5);
  }
#undef TTK_PROBE_U
#undef TTK_PROBE
  TTK_LAUNCH_CHECK("ttk_stream_probe");
}
