// Multi-task heads of NetworkWithPointHead (neuralnets/models.py:340-376 after the backbone):
// one stacked linear layer z = Wcat.f + bcat followed by the per-sample head arithmetic of
// head_math.h (box, position/size, quaternion, triangular uncertainty scales, local pose offsets,
// 3DMM landmarks).  One wavefront per sample; dot products and the 68-landmark sums are reduced with
// wave shuffles.  Tiny next to the backbone (0.1 % of the FLOPs): built for few launches, not for MFMA.
#include "head_math.h"
#include "ttk_common.h"
#include <stdlib.h>

namespace ttk {

using hm::Q;
constexpr int kMaxZ = 80;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

struct HeadsArgs {
  const float *feat, *wcat, *bcat, *P, *Pk, *kp, *eig;
  const int* ids;
  int B, F, NZ, unc, pt, use_offset, rot6d;
};

// P = keypt + sum_i eig_i * shapeparam_i of landmark p (DeformableHeadKeypoints, modelcomponents.py:59-82)
__device__ __forceinline__ void landmark_local(const HeadsArgs& a, const float* sh, int p, float local[3]) {
  local[0] = a.kp[p * 3]; local[1] = a.kp[p * 3 + 1]; local[2] = a.kp[p * 3 + 2];
  for (int i = 0; i < 50; ++i) {
    const float* e = a.eig + ((size_t)i * 68 + p) * 3;
    const float c = sh[i];
    local[0] = fmaf(e[0], c, local[0]); local[1] = fmaf(e[1], c, local[1]); local[2] = fmaf(e[2], c, local[2]);
  }
}

__global__ void __launch_bounds__(kBlock) heads_fwd_k(HeadsArgs a, float* __restrict__ z, float* __restrict__ roi,
                                                       float* __restrict__ coord, float* __restrict__ rot,
                                                       float* __restrict__ qu, float* __restrict__ Lc,
                                                       float* __restrict__ Lr, float* __restrict__ pts,
                                                       float* __restrict__ shp) {
  // One workgroup per sample: its four waves share the rows of the stacked linear layer (the kernel is a chain of
  // load latencies - a wave per sample left half the CUs idle and took 56 us at B = 512), then wave 0 finishes the sample.
  __shared__ float zs1[kMaxZ];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = blockIdx.x;
  const float* f = a.feat + (size_t)s * a.F;
  // four rows at a time: their weight loads are independent, so a wave has 4x the loads in flight of a row-by-row loop
  for (int j0 = 4 * wv; j0 < a.NZ; j0 += 4 * (kBlock / kWave)) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane * 4; k < a.F; k += 256) {
      const float4 fv = ld4(f + k);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + u, a.NZ - 1);
        const float4 wv4 = ld4(a.wcat + (size_t)j * a.F + k);
        acc[u] = fmaf(fv.x, wv4.x, fmaf(fv.y, wv4.y, fmaf(fv.z, wv4.z, fmaf(fv.w, wv4.w, acc[u]))));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v = wave_sum(acc[u]);
      const int j = j0 + u;
      if (lane == 0 && j < a.NZ) {
        const float zz = v + a.bcat[j];
        zs1[j] = zz;
        z[(size_t)s * a.NZ + j] = zz;
      }
    }
  }
  __syncthreads();
  if (wv != 0) return;
  const float* zz = zs1;
  const int id = a.ids ? a.ids[s] : 0;
  const float* prow = a.P ? a.P + 4 * id : nullptr;
  const float* pkrow = a.Pk ? a.Pk + 4 * id : nullptr;
  const float* sh = zz + hm::z_shape(a.unc, a.rot6d);
  if (a.pt && lane < 50) shp[50 * s + lane] = sh[lane];
  if (!a.rot6d) {
    hm::HeadOut o;
    hm::sample_fwd_core(zz, a.unc, a.pt, a.use_offset, prow, pkrow, o);
    if (lane == 0) {
      for (int i = 0; i < 4; ++i) roi[4 * s + i] = o.roi[i];
      for (int i = 0; i < 3; ++i) coord[3 * s + i] = o.coord[i];
      rot[4 * s] = o.rot.i; rot[4 * s + 1] = o.rot.j; rot[4 * s + 2] = o.rot.k; rot[4 * s + 3] = o.rot.w;
      qu[4 * s] = o.qu.i; qu[4 * s + 1] = o.qu.j; qu[4 * s + 2] = o.qu.k; qu[4 * s + 3] = o.qu.w;
      if (a.unc)
        for (int i = 0; i < 9; ++i) { Lc[9 * s + i] = o.Lc[i]; Lr[9 * s + i] = o.Lr[i]; }
    }
    if (a.pt)
      for (int p = lane; p < 68; p += 64) {
        float local[3], out[3];
        landmark_local(a, sh, p, local);
        hm::landmark_fwd(o.qk, o.ck, local, out);
        float* d = pts + ((size_t)s * 68 + p) * 3;
        d[0] = out[0]; d[1] = out[1]; d[2] = out[2];
      }
  } else {  // 6D rotation head: rot[B][9] row-major matrices, qu[B][6] = the raw 6D features
    hm::HeadOutM o;
    hm::sample_fwd_core_m(zz, a.unc, a.pt, a.use_offset, prow, pkrow, o);
    if (lane == 0) {
      for (int i = 0; i < 4; ++i) roi[4 * s + i] = o.roi[i];
      for (int i = 0; i < 3; ++i) coord[3 * s + i] = o.coord[i];
      for (int i = 0; i < 9; ++i) rot[9 * s + i] = o.rot[i];
      for (int i = 0; i < 6; ++i) qu[6 * s + i] = zz[hm::Z_QUAT + i];
      if (a.unc)
        for (int i = 0; i < 9; ++i) { Lc[9 * s + i] = o.Lc[i]; Lr[9 * s + i] = o.Lr[i]; }
    }
    if (a.pt)
      for (int p = lane; p < 68; p += 64) {
        float local[3], out[3];
        landmark_local(a, sh, p, local);
        hm::landmark_fwd_m(o.Rk, o.ck, local, out);
        float* d = pts + ((size_t)s * 68 + p) * 3;
        d[0] = out[0]; d[1] = out[1]; d[2] = out[2];
      }
  }
}

struct HeadsGradIn {
  const float *roi, *coord, *rot, *qu, *Lc, *Lr, *pts, *shp;
};

// per-sample backward: dz[B][NZ], dprow[B][8] (gradients w.r.t. this sample's rows of p and p_kpts)
// shared tail of the per-sample backward: d shapeparam_i = sum_p <glocal_p, eig_i,p> + direct gradient
__device__ __forceinline__ void shape_grad(const HeadsArgs& a, const float (*gl)[3], const float* g_shp, int s, int lane, float* dzs,
                                           int zshape) {
  // lane i owns shape parameter i: the landmark gradients go through a wave-private LDS tile and every lane runs its own
  // 204-term dot product with 16-byte loads (50 wave-wide reductions in a row made this kernel a 40 us latency chain)
  __shared__ __attribute__((aligned(16))) float gls[kBlock / kWave][68 * 3];
  float* mine = gls[threadIdx.x >> 6];
  int q = 0;
  for (int pnt = lane; pnt < 68; pnt += 64, ++q) {
    mine[3 * pnt] = gl[q][0]; mine[3 * pnt + 1] = gl[q][1]; mine[3 * pnt + 2] = gl[q][2];
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  if (lane < 50) {
    const float* e = a.eig + (size_t)lane * 204;
    float acc = 0.f;
#pragma unroll 17
    for (int k = 0; k < 204; k += 4) {
      const float4 ev = ld4(e + k), gv = ld4(mine + k);
      acc = fmaf(ev.x, gv.x, fmaf(ev.y, gv.y, fmaf(ev.z, gv.z, fmaf(ev.w, gv.w, acc))));
    }
    dzs[zshape + lane] = acc + g_shp[50 * s + lane];
  }
}

__global__ void __launch_bounds__(kBlock) heads_bwd_sample_k(HeadsArgs a, const float* __restrict__ z, HeadsGradIn g,
                                                              float* __restrict__ dz, float* __restrict__ dprow) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = blockIdx.x * (kBlock / kWave) + wv;
  if (s >= a.B) return;
  const float* zz = z + (size_t)s * a.NZ;
  const int id = a.ids ? a.ids[s] : 0;
  const float* p = a.P ? a.P + 4 * id : nullptr;
  const float* pk = a.Pk ? a.Pk + 4 * id : nullptr;
  float* dzs = dz + (size_t)s * a.NZ;
  const int zshape = hm::z_shape(a.unc, a.rot6d);
  const float* sh = zz + zshape;
  float gl[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  float gz[hm::Z_BASE + 2 + 14];
  float gp[4] = {0.f, 0.f, 0.f, 0.f}, gpk[4] = {0.f, 0.f, 0.f, 0.f};
  if (!a.rot6d) {
    hm::HeadOut o;
    hm::sample_fwd_core(zz, a.unc, a.pt, a.use_offset, p, pk, o);
    hm::HeadGrad hg;
    for (int i = 0; i < 4; ++i) hg.roi[i] = g.roi[4 * s + i];
    for (int i = 0; i < 3; ++i) { hg.coord[i] = g.coord[3 * s + i]; hg.ck[i] = 0.f; }
    hg.rot = Q{g.rot[4 * s], g.rot[4 * s + 1], g.rot[4 * s + 2], g.rot[4 * s + 3]};
    hg.qu = Q{g.qu[4 * s], g.qu[4 * s + 1], g.qu[4 * s + 2], g.qu[4 * s + 3]};
    hg.qk = Q{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 9; ++i) { hg.Lc[i] = a.unc ? g.Lc[9 * s + i] : 0.f; hg.Lr[i] = a.unc ? g.Lr[9 * s + i] : 0.f; }
    if (a.pt) {
      Q gqk{0.f, 0.f, 0.f, 0.f};
      float gck[3] = {0.f, 0.f, 0.f};
      int np = 0;
      for (int pnt = lane; pnt < 68; pnt += 64, ++np) {
        float local[3];
        landmark_local(a, sh, pnt, local);
        hm::landmark_bwd(o.qk, o.ck, local, g.pts + ((size_t)s * 68 + pnt) * 3, gqk, gck, gl[np]);
      }
      shape_grad(a, gl, g.shp, s, lane, dzs, zshape);
      hg.qk = Q{wave_sum(gqk.i), wave_sum(gqk.j), wave_sum(gqk.k), wave_sum(gqk.w)};
      for (int i = 0; i < 3; ++i) hg.ck[i] = wave_sum(gck[i]);
    }
    if (lane == 0) hm::sample_bwd_core(zz, a.unc, a.pt, a.use_offset, p, pk, hg, gz, gp, gpk);
  } else {
    hm::HeadOutM o;
    hm::sample_fwd_core_m(zz, a.unc, a.pt, a.use_offset, p, pk, o);
    hm::HeadGradM hg;
    for (int i = 0; i < 4; ++i) hg.roi[i] = g.roi[4 * s + i];
    for (int i = 0; i < 3; ++i) { hg.coord[i] = g.coord[3 * s + i]; hg.ck[i] = 0.f; }
    for (int i = 0; i < 9; ++i) { hg.rot[i] = g.rot[9 * s + i]; hg.Rk[i] = 0.f; }
    for (int i = 0; i < 6; ++i) hg.z6[i] = g.qu[6 * s + i];
    for (int i = 0; i < 9; ++i) { hg.Lc[i] = a.unc ? g.Lc[9 * s + i] : 0.f; hg.Lr[i] = a.unc ? g.Lr[9 * s + i] : 0.f; }
    if (a.pt) {
      float gRk[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float gck[3] = {0.f, 0.f, 0.f};
      int np = 0;
      for (int pnt = lane; pnt < 68; pnt += 64, ++np) {
        float local[3];
        landmark_local(a, sh, pnt, local);
        hm::landmark_bwd_m(o.Rk, o.ck, local, g.pts + ((size_t)s * 68 + pnt) * 3, gRk, gck, gl[np]);
      }
      shape_grad(a, gl, g.shp, s, lane, dzs, zshape);
      for (int i = 0; i < 9; ++i) hg.Rk[i] = wave_sum(gRk[i]);
      for (int i = 0; i < 3; ++i) hg.ck[i] = wave_sum(gck[i]);
    }
    if (lane == 0) hm::sample_bwd_core_m(zz, a.unc, a.pt, a.use_offset, p, pk, hg, gz, gp, gpk);
  }
  if (lane == 0) {
    for (int i = 0; i < zshape; ++i) dzs[i] = gz[i];
    for (int i = 0; i < 4; ++i) { dprow[8 * s + i] = gp[i]; dprow[8 * s + 4 + i] = gpk[i]; }
  }
}

// dfeat[b][f] = sum_j dz[b][j] * W[j][f]; the same threads zero dW / db for the weight kernel that follows (it adds atomically:
// two fill launches less)
__global__ void __launch_bounds__(kBlock) heads_bwd_feat_k(const float* __restrict__ dz, const float* __restrict__ wcat,
                                                            float* __restrict__ dfeat, float* __restrict__ dw_zero, float* __restrict__ db_zero,
                                                            int B, int F, int NZ) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int fq = F >> 2;
  if (idx < (int64_t)NZ * fq) st4(dw_zero + 4 * idx, f4(0.f));
  if (idx < NZ) db_zero[idx] = 0.f;
  if (idx >= (int64_t)B * fq) return;
  const int b = (int)(idx / fq), f = (int)(idx % fq) * 4;
  float4 acc = f4(0.f);
  for (int j = 0; j < NZ; ++j) acc = fma4(f4(dz[(size_t)b * NZ + j]), ld4(wcat + (size_t)j * F + f), acc);
  st4(dfeat + (size_t)b * F + f, acc);
}

// dW[j][f] += sum_{b in chunk} dz[b][j]*feat[b][f];  db[j] += sum_b dz[b][j]   (chunks of 64 samples, atomics)
__global__ void __launch_bounds__(kBlock) heads_bwd_weight_k(const float* __restrict__ dz, const float* __restrict__ feat,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int F,
                                                              int NZ) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int fq = F >> 2;
  if (idx >= (int64_t)NZ * fq) return;
  const int j = (int)(idx / fq), f = (int)(idx % fq) * 4;
  const int chunk = gridDim.y == 1 ? B : 64;  // one chunk (deterministic mode): plain stores of the complete sums
  const int b0 = blockIdx.y * chunk, b1 = min(B, b0 + chunk);
  float4 acc = f4(0.f);
  float sb = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float d = dz[(size_t)b * NZ + j];
    acc = fma4(f4(d), ld4(feat + (size_t)b * F + f), acc);
    sb += d;
  }
  float* o = dw + (size_t)j * F + f;
  atomicAdd(o, acc.x); atomicAdd(o + 1, acc.y); atomicAdd(o + 2, acc.z); atomicAdd(o + 3, acc.w);
  if (f == 0) atomicAdd(db + j, sb);
}

// dP[r][c] = sum_{b: id_b == r} dprow[b][c] (c<4), dPk likewise (c>=4).  One workgroup per output element
// (module, row, component), its threads stride over the samples; fixed-order reduction.
__global__ void __launch_bounds__(kBlock) heads_bwd_offset_k(const float* __restrict__ dprow, const int* __restrict__ ids,
                                                              float* __restrict__ dP, float* __restrict__ dPk, int B) {
  __shared__ float red[kBlock];
  const int t = blockIdx.x;  // 0..63: module (1 bit), row (3 bits), component (2 bits)
  const int mod = t >> 5, r = (t >> 2) & 7, c = t & 3;
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += kBlock)
    if ((ids ? ids[b] : 0) == r) acc += dprow[8 * b + 4 * mod + c];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  float* out = mod ? dPk : dP;
  if (threadIdx.x == 0 && out) out[4 * r + c] = red[0];
}

// DiagonalScaleParameter (negloglikelihood.py:50-65): out_i = elu1(h0)*elu1(h_{1+i}) + 1e-6
__global__ void diag_scale_fwd_k(const float* h, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = hm::elu1(h[0]) * hm::elu1(h[1 + i]) + 1.0e-6f;
}
__global__ void diag_scale_bwd_k(const float* h, const float* g, float* gh, int n) {  // one block
  __shared__ float red[256];
  float acc = 0.f;
  const float m = hm::elu1(h[0]);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    acc += g[i] * hm::elu1(h[1 + i]);
    gh[1 + i] = g[i] * m * hm::elu1_d(h[1 + i]);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)blockDim.x; ++i) s += red[i];
    gh[0] = s * hm::elu1_d(h[0]);
  }
}

}  // namespace ttk

using namespace ttk;

static int heads_check(const char* name, int B, int F, int NZ, int unc, int pt, int rot6d) {
  if (B <= 0 || F <= 0 || (F & 3) || NZ != hm::z_count(unc, pt, rot6d) || NZ > kMaxZ) {
    set_error("%s: bad sizes B=%d F=%d (NZ=%d expected %d)", name, B, F, NZ, hm::z_count(unc, pt, rot6d));
    return -1;
  }
  return 0;
}

extern "C" {

int ttk_heads_num_rows(int enable_uncertainty, int enable_point_head, int enable_6drot) {
  return hm::z_count(enable_uncertainty, enable_point_head, enable_6drot);
}

int ttk_heads_fwd(const float* feat, const float* wcat, const float* bcat, const int* ids, const float* P, const float* Pk,
                  const float* keypts, const float* keyeig, int B, int F, int NZ, int enable_uncertainty, int enable_point_head,
                  int use_offset, int enable_6drot, float* z, float* roi, float* coord, float* rot, float* qu, float* Lc, float* Lr,
                  float* pts, float* shp, ttk_stream_t stream) {
  if (heads_check("heads_fwd", B, F, NZ, enable_uncertainty, enable_point_head, enable_6drot)) return -1;
  TTK_REQUIRE(feat && wcat && bcat && z && roi && coord && rot && qu, "heads_fwd: null pointer");
  TTK_REQUIRE(!enable_uncertainty || (Lc && Lr), "heads_fwd: uncertainty outputs missing");
  TTK_REQUIRE(!enable_point_head || (pts && shp && keypts && keyeig), "heads_fwd: point-head buffers missing");
  TTK_REQUIRE(!use_offset || (P && (!enable_point_head || Pk)), "heads_fwd: local pose offset parameters missing");
  HeadsArgs a{feat, wcat, bcat, P, Pk, keypts, keyeig, ids, B, F, NZ, enable_uncertainty, enable_point_head, use_offset, enable_6drot};
  hipLaunchKernelGGL(heads_fwd_k, dim3(B), dim3(kBlock), 0, (hipStream_t)stream, a, z, roi, coord, rot, qu, Lc, Lr,
                     pts, shp);
  TTK_LAUNCH_CHECK("heads_fwd");
}

int ttk_heads_bwd(const float* feat, const float* wcat, const float* z, const int* ids, const float* P, const float* Pk,
                  const float* keypts, const float* keyeig, int B, int F, int NZ, int enable_uncertainty, int enable_point_head,
                  int use_offset, int enable_6drot, const float* g_roi, const float* g_coord, const float* g_rot, const float* g_qu,
                  const float* g_Lc, const float* g_Lr, const float* g_pts, const float* g_shp, float* dz, float* dprow,
                  float* dfeat, float* dwcat, float* dbcat, float* dP, float* dPk, ttk_stream_t stream) {
  if (heads_check("heads_bwd", B, F, NZ, enable_uncertainty, enable_point_head, enable_6drot)) return -1;
  TTK_REQUIRE(feat && wcat && z && g_roi && g_coord && g_rot && g_qu && dz && dprow && dfeat && dwcat && dbcat, "heads_bwd: null pointer");
  TTK_REQUIRE(!enable_uncertainty || (g_Lc && g_Lr), "heads_bwd: uncertainty gradients missing");
  TTK_REQUIRE(!enable_point_head || (g_pts && g_shp && keypts && keyeig), "heads_bwd: point-head gradients missing");
  hipStream_t st = (hipStream_t)stream;
  HeadsArgs a{feat, wcat, nullptr, P, Pk, keypts, keyeig, ids, B, F, NZ, enable_uncertainty, enable_point_head, use_offset, enable_6drot};
  HeadsGradIn g{g_roi, g_coord, g_rot, g_qu, g_Lc, g_Lr, g_pts, g_shp};
  hipLaunchKernelGGL(heads_bwd_sample_k, dim3((B + 3) / 4), dim3(kBlock), 0, st, a, z, g, dz, dprow);
  const int64_t nf = (int64_t)B * (F / 4), nw = (int64_t)NZ * (F / 4);
  hipLaunchKernelGGL(heads_bwd_feat_k, dim3((unsigned)ceil_div(nf > nw ? nf : nw, kBlock)), dim3(kBlock), 0, st, dz, wcat, dfeat, dwcat, dbcat, B, F,
                     NZ);
  // TTK_DETERMINISTIC=1: one chunk of samples per weight instead of B/64 chunks that add atomically (fixed summation order)
  const bool det = deterministic_mode();
  hipLaunchKernelGGL(heads_bwd_weight_k, dim3((unsigned)ceil_div(nw, kBlock), det ? 1u : (unsigned)ceil_div(B, 64)), dim3(kBlock), 0, st, dz,
                     feat, dwcat, dbcat, B, F, NZ);
  if (use_offset && (dP || dPk)) hipLaunchKernelGGL(heads_bwd_offset_k, dim3(64), dim3(kBlock), 0, st, dprow, ids, dP, dPk, B);
  TTK_LAUNCH_CHECK("heads_bwd");
}

int ttk_diag_scale_fwd(const float* hidden, float* out, int n, ttk_stream_t stream) {
  TTK_REQUIRE(hidden && out && n > 0, "diag_scale_fwd: bad arguments");
  hipLaunchKernelGGL(diag_scale_fwd_k, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, hidden, out, n);
  TTK_LAUNCH_CHECK("diag_scale_fwd");
}
int ttk_diag_scale_bwd(const float* hidden, const float* gout, float* ghidden, int n, ttk_stream_t stream) {
  TTK_REQUIRE(hidden && gout && ghidden && n > 0, "diag_scale_bwd: bad arguments");
  hipLaunchKernelGGL(diag_scale_bwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, hidden, gout, ghidden, n);
  TTK_LAUNCH_CHECK("diag_scale_bwd");
}

}  // extern "C"
