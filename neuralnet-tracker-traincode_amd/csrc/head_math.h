// Per-sample arithmetic of the multi-task heads, forward and hand-derived backward.
// Host + device header: compiled by hipcc into heads.hip, and by g++ into a test-only harness
// (tests/host_math) that checks every formula against autograd of the CPU oracle.
//
// Reference (paths relative to trackertraincode/):
//   neuralnets/models.py:127-150,177-215      BoundingBox / PositionSizeOutput / DirectQuaternionWithNormalization
//   neuralnets/rotrepr.py:36-48               QuatRepr.from_features (elu+1 on w, L2-normalise eps 1e-6)
//   neuralnets/torchquaternion.py:23-67       mult / rotate (quaternion order i,j,k,w)
//   neuralnets/negloglikelihood.py:22-35,187-242   Neck + FeaturesAsTriangularScale
//   neuralnets/modelcomponents.py:38-82,136-184    rigid_transformation_25d, DeformableHeadKeypoints,
//                                                  LocalToGlobalCoordinateOffset
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TTK_HD __host__ __device__ __forceinline__
#else
#define TTK_HD inline
#endif

namespace ttk {
namespace hm {

// Row layout of the stacked linear layer z = Wcat.f + bcat
enum : int {
  Z_BOX = 0,    // 4: boxnet.linear
  Z_XY = 4,     // 2: posnet.linear_xy
  Z_SIZE = 6,   // 1: posnet.linear_size
  Z_QUAT = 7,   // 4: quatnet.linear (DirectQuaternionWithNormalization) or 6 (RotRepr6dWithNormalization)
  Z_BASE = 11,  // rows present in every quaternion configuration (13 with the 6D head)
  // with uncertainty: +7 posnet.scales.neck.lin, +7 quatnet.uncertainty_net.neck.lin
  // with point head : +50 landmarks.shapenet (after the uncertainty rows)
};
TTK_HD int z_base(bool rot6d) { return rot6d ? Z_BASE + 2 : Z_BASE; }
TTK_HD int z_coord_scale(bool, bool rot6d = false) { return z_base(rot6d); }
TTK_HD int z_pose_scale(bool, bool rot6d = false) { return z_base(rot6d) + 7; }
TTK_HD int z_shape(bool unc, bool rot6d = false) { return z_base(rot6d) + (unc ? 14 : 0); }
TTK_HD int z_count(bool unc, bool pt, bool rot6d = false) { return z_base(rot6d) + (unc ? 14 : 0) + (pt ? 50 : 0); }

struct Q {
  float i, j, k, w;
};
TTK_HD Q qmul(Q u, Q v) {  // Hamilton product, torchquaternion.py:23-48
  Q r;
  r.i = u.i * v.w + u.w * v.i - u.k * v.j + u.j * v.k;
  r.j = u.j * v.w + u.k * v.i + u.w * v.j - u.i * v.k;
  r.k = u.k * v.w - u.j * v.i + u.i * v.j + u.w * v.k;
  r.w = u.w * v.w - u.i * v.i - u.j * v.j - u.k * v.k;
  return r;
}
TTK_HD Q qconj(Q q) { return Q{-q.i, -q.j, -q.k, q.w}; }
TTK_HD Q qadd(Q a, Q b) { return Q{a.i + b.i, a.j + b.j, a.k + b.k, a.w + b.w}; }
TTK_HD float qdot(Q a, Q b) { return a.i * b.i + a.j * b.j + a.k * b.k + a.w * b.w; }

// smoothclip0 = elu(x) + 1 (neuralnets/math.py:34-37) and its derivative
TTK_HD float elu1(float x) { return x > 0.f ? x + 1.f : expf(x); }
TTK_HD float elu1_d(float x) { return x > 0.f ? 1.f : expf(x); }

// (q (t,0) q*)_ijk - torchquaternion.py:51-67 (NOT the unit-quaternion shortcut: scales with |q|^2)
TTK_HD void qrot(Q q, const float t[3], float r[3]) {
  const Q u = qmul(qmul(q, Q{t[0], t[1], t[2], 0.f}), qconj(q));
  r[0] = u.i; r[1] = u.j; r[2] = u.k;
}
// Adjoint identities of the quaternion product c = a*b under <x,y> = sum of components:
//   dL/da = g * conj(b),  dL/db = conj(a) * g.
TTK_HD void qrot_bwd(Q q, const float t[3], const float gr[3], Q& gq, float gt[3]) {
  const Q T{t[0], t[1], t[2], 0.f};
  const Q u = qmul(q, T);
  const Q g4{gr[0], gr[1], gr[2], 0.f};
  const Q gu = qmul(g4, q);             // r4 = u * conj(q): dL/du = g4 * conj(conj(q))
  const Q gqc = qmul(qconj(u), g4);     // dL/d conj(q)
  gq = qadd(gq, qconj(gqc));
  gq = qadd(gq, qmul(gu, qconj(T)));    // u = q * T
  const Q gT = qmul(qconj(q), gu);
  gt[0] += gT.i; gt[1] += gT.j; gt[2] += gT.k;
}

// ---- QuatRepr.from_features: qu = [z_ijk, elu1(z_w)], q = qu / max(|qu|, 1e-6) -------------------
TTK_HD void quat_head_fwd(const float z[4], Q& qu, Q& q) {
  qu = Q{z[0], z[1], z[2], elu1(z[3])};
  const float n = fmaxf(sqrtf(qdot(qu, qu)), 1.0e-6f);
  const float inv = 1.f / n;
  q = Q{qu.i * inv, qu.j * inv, qu.k * inv, qu.w * inv};
}
// gq: grad w.r.t. normalised q; gqu_direct: grad w.r.t. the `unnormalized_quat` output
TTK_HD void quat_head_bwd(const float z[4], Q gq, Q gqu_direct, float gz[4]) {
  Q qu, q;
  quat_head_fwd(z, qu, q);
  const float nrm = sqrtf(qdot(qu, qu));
  Q g;
  if (nrm > 1.0e-6f) {
    const float d = qdot(q, gq), inv = 1.f / nrm;
    g = Q{(gq.i - q.i * d) * inv, (gq.j - q.j * d) * inv, (gq.k - q.k * d) * inv, (gq.w - q.w * d) * inv};
  } else {
    g = Q{gq.i * 1.0e6f, gq.j * 1.0e6f, gq.k * 1.0e6f, gq.w * 1.0e6f};
  }
  g = qadd(g, gqu_direct);
  gz[0] = g.i; gz[1] = g.j; gz[2] = g.k; gz[3] = g.w * elu1_d(z[3]);
}

// ---- FeaturesAsTriangularScale: x[7] -> lower-triangular L (row-major 3x3) -------------------------
TTK_HD void tri_scale_fwd(const float x[7], float L[9]) {
  const float m = elu1(x[0]);
  const float z0 = m * elu1(x[1]) + 1.0e-6f, z1 = m * elu1(x[2]) + 1.0e-6f, z2 = m * elu1(x[3]) + 1.0e-6f;
  const float z3 = m * x[4], z4 = m * x[5], z5 = m * x[6];
  L[0] = z0; L[1] = 0.f; L[2] = 0.f;
  L[3] = z3; L[4] = z1; L[5] = 0.f;
  L[6] = z4; L[7] = z5; L[8] = z2;
}
TTK_HD void tri_scale_bwd(const float x[7], const float gL[9], float gx[7]) {
  const float m = elu1(x[0]);
  const float gz0 = gL[0], gz1 = gL[4], gz2 = gL[8], gz3 = gL[3], gz4 = gL[6], gz5 = gL[7];
  const float gm = gz0 * elu1(x[1]) + gz1 * elu1(x[2]) + gz2 * elu1(x[3]) + gz3 * x[4] + gz4 * x[5] + gz5 * x[6];
  gx[0] = gm * elu1_d(x[0]);
  gx[1] = gz0 * m * elu1_d(x[1]);
  gx[2] = gz1 * m * elu1_d(x[2]);
  gx[3] = gz2 * m * elu1_d(x[3]);
  gx[4] = gz3 * m; gx[5] = gz4 * m; gx[6] = gz5 * m;
}

// ---- BoundingBox: z[4] -> roi = [c - s, c + s], s = elu1(z[2:4]) ------------------------------------
TTK_HD void box_fwd(const float z[4], float roi[4]) {
  const float s0 = elu1(z[2]), s1 = elu1(z[3]);
  roi[0] = z[0] - s0; roi[1] = z[1] - s1; roi[2] = z[0] + s0; roi[3] = z[1] + s1;
}
TTK_HD void box_bwd(const float z[4], const float g[4], float gz[4]) {
  gz[0] = g[0] + g[2];
  gz[1] = g[1] + g[3];
  gz[2] = (g[2] - g[0]) * elu1_d(z[2]);
  gz[3] = (g[3] - g[1]) * elu1_d(z[3]);
}

// ---- LocalToGlobalCoordinateOffset (modelcomponents.py:136-184) -----------------------------------
// Quirk kept: p[1] is BOTH the x-rotation angle and the first translation component; p[0] unused.
TTK_HD void offset_fwd(const float p[4], Q hq, const float hc[3], Q& q, float c[3]) {
  const float h = 0.5f * p[1];
  const Q qo{sinf(h), 0.f, 0.f, cosf(h)};
  const float t[3] = {0.f, p[1], p[2]};
  const float size = hc[2] * elu1(p[3]);
  q = qmul(hq, qo);
  float r[3];
  qrot(hq, t, r);
  c[0] = hc[0] + r[0] * size;
  c[1] = hc[1] + r[1] * size;
  c[2] = size;
}
// accumulates into ghq, ghc[3], gp[4]
TTK_HD void offset_bwd(const float p[4], Q hq, const float hc[3], Q gq, const float gc[3], Q& ghq, float ghc[3],
                       float gp[4]) {
  const float h = 0.5f * p[1], sh = sinf(h), ch = cosf(h);
  const Q qo{sh, 0.f, 0.f, ch};
  const float t[3] = {0.f, p[1], p[2]};
  const float so = elu1(p[3]);
  const float size = hc[2] * so;
  float r[3];
  qrot(hq, t, r);
  const float gsize = gc[2] + gc[0] * r[0] + gc[1] * r[1];
  const float gr[3] = {gc[0] * size, gc[1] * size, 0.f};
  ghc[0] += gc[0];
  ghc[1] += gc[1];
  ghc[2] += gsize * so;
  gp[3] += gsize * hc[2] * elu1_d(p[3]);
  ghq = qadd(ghq, qmul(gq, qconj(qo)));
  const Q gqo = qmul(qconj(hq), gq);
  gp[1] += 0.5f * (gqo.i * ch - gqo.w * sh);
  float gt[3] = {0.f, 0.f, 0.f};
  qrot_bwd(hq, t, gr, ghq, gt);
  gp[1] += gt[1];
  gp[2] += gt[2];
}

// ---- one landmark: P = keypt + sum_i eig_i*shp_i ; out = rot(qk, P)*size ; out_xy += xy -------------
TTK_HD void landmark_fwd(Q qk, const float ck[3], const float local[3], float out[3]) {
  float r[3];
  qrot(qk, local, r);
  out[0] = r[0] * ck[2] + ck[0];
  out[1] = r[1] * ck[2] + ck[1];
  out[2] = r[2] * ck[2];
}
// accumulates gqk, gck[3]; writes glocal[3]
TTK_HD void landmark_bwd(Q qk, const float ck[3], const float local[3], const float g[3], Q& gqk, float gck[3],
                         float glocal[3]) {
  float r[3];
  qrot(qk, local, r);
  gck[0] += g[0];
  gck[1] += g[1];
  gck[2] += g[0] * r[0] + g[1] * r[1] + g[2] * r[2];
  const float gr[3] = {g[0] * ck[2], g[1] * ck[2], g[2] * ck[2]};
  glocal[0] = glocal[1] = glocal[2] = 0.f;
  qrot_bwd(qk, local, gr, gqk, glocal);
}

// ---- everything of one sample except the 68-landmark loop ------------------------------------------
struct HeadOut {
  float roi[4], coord[3], Lc[9], Lr[9], ck[3];
  Q rot, qu, qk;
};
// p / pk: rows of local_pose_offset.p / local_pose_offset_kpts.p for this sample (ignored if !use_offset)
TTK_HD void sample_fwd_core(const float* z, bool unc, bool pt, bool use_offset, const float* p, const float* pk,
                            HeadOut& o) {
  box_fwd(z + Z_BOX, o.roi);
  const float hc[3] = {z[Z_XY], z[Z_XY + 1], elu1(z[Z_SIZE])};
  Q hq;
  quat_head_fwd(z + Z_QUAT, o.qu, hq);
  if (unc) {
    tri_scale_fwd(z + z_coord_scale(unc), o.Lc);
    tri_scale_fwd(z + z_pose_scale(unc), o.Lr);
  }
  if (use_offset) {
    offset_fwd(p, hq, hc, o.rot, o.coord);
    if (pt) offset_fwd(pk, hq, hc, o.qk, o.ck);
  } else {
    o.rot = hq;
    o.coord[0] = hc[0]; o.coord[1] = hc[1]; o.coord[2] = hc[2];
    o.qk = hq;
    o.ck[0] = hc[0]; o.ck[1] = hc[1]; o.ck[2] = hc[2];
  }
}
// Upstream gradients of one sample; qk/ck are produced by the landmark loop (zero without point head).
struct HeadGrad {
  float roi[4], coord[3], Lc[9], Lr[9], ck[3];
  Q rot, qu, qk;
};
// Writes gz[0 .. z_shape(unc)) (the shape rows come from the landmark loop); accumulates gp[4], gpk[4].
TTK_HD void sample_bwd_core(const float* z, bool unc, bool pt, bool use_offset, const float* p, const float* pk,
                            const HeadGrad& g, float* gz, float gp[4], float gpk[4]) {
  box_bwd(z + Z_BOX, g.roi, gz + Z_BOX);
  const float hc[3] = {z[Z_XY], z[Z_XY + 1], elu1(z[Z_SIZE])};
  Q qu, hq;
  quat_head_fwd(z + Z_QUAT, qu, hq);
  Q ghq{0.f, 0.f, 0.f, 0.f};
  float ghc[3] = {0.f, 0.f, 0.f};
  if (use_offset) {
    offset_bwd(p, hq, hc, g.rot, g.coord, ghq, ghc, gp);
    if (pt) offset_bwd(pk, hq, hc, g.qk, g.ck, ghq, ghc, gpk);
  } else {
    ghq = qadd(g.rot, g.qk);
    for (int i = 0; i < 3; ++i) ghc[i] = g.coord[i] + g.ck[i];
  }
  gz[Z_XY] = ghc[0];
  gz[Z_XY + 1] = ghc[1];
  gz[Z_SIZE] = ghc[2] * elu1_d(z[Z_SIZE]);
  quat_head_bwd(z + Z_QUAT, ghq, g.qu, gz + Z_QUAT);
  if (unc) {
    tri_scale_bwd(z + z_coord_scale(unc), g.Lc, gz + z_coord_scale(unc));
    tri_scale_bwd(z + z_pose_scale(unc), g.Lr, gz + z_pose_scale(unc));
  }
}

// =====================================================================================================
// 6D rotation head (RotRepr6dWithNormalization, models.py:153-174): rotations are 3x3 matrices (row-major
// float[9]) instead of quaternions.  neuralnets/torch6drotation.py:27-49 (tomatrix), rotrepr.py:63-98 (Mat33Repr).
// =====================================================================================================
TTK_HD void v3cross(const float a[3], const float b[3], float r[3]) {
  r[0] = a[1] * b[2] - a[2] * b[1];
  r[1] = a[2] * b[0] - a[0] * b[2];
  r[2] = a[0] * b[1] - a[1] * b[0];
}
TTK_HD float v3dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// F.normalize(v, eps=1e-6): v / max(|v|, eps)
TTK_HD void v3normalize(const float v[3], float u[3]) {
  const float inv = 1.f / fmaxf(sqrtf(v3dot(v, v)), 1.0e-6f);
  u[0] = v[0] * inv; u[1] = v[1] * inv; u[2] = v[2] * inv;
}
TTK_HD void v3normalize_bwd(const float v[3], const float g[3], float gv[3]) {
  const float n = sqrtf(v3dot(v, v));
  if (n > 1.0e-6f) {
    const float inv = 1.f / n, u[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
    const float d = v3dot(u, g);
    for (int i = 0; i < 3; ++i) gv[i] = (g[i] - u[i] * d) * inv;
  } else {
    for (int i = 0; i < 3; ++i) gv[i] = g[i] * 1.0e6f;
  }
}
TTK_HD void m3mul(const float a[9], const float b[9], float c[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
TTK_HD void m3vec(const float a[9], const float v[3], float r[3]) {
  for (int i = 0; i < 3; ++i) r[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}
TTK_HD void m3tvec(const float a[9], const float v[3], float r[3]) {  // a^T v
  for (int j = 0; j < 3; ++j) r[j] = a[j] * v[0] + a[3 + j] * v[1] + a[6 + j] * v[2];
}

// z[6] -> R: rows x/|x|, ((x X y) X x)/|.|, (x X y)/|.|; identity if max|R R^T - I| > 1e-3.  Returns whether the
// identity fallback was taken (then no gradient flows to z through R).
TTK_HD bool rot6d_fwd(const float z[6], float R[9]) {
  const float* x = z;
  const float* y = z + 3;
  float c[3], y2[3];
  v3cross(x, y, c);
  v3cross(c, x, y2);
  v3normalize(x, R);
  v3normalize(y2, R + 3);
  v3normalize(c, R + 6);
  float bad = 0.f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) bad = fmaxf(bad, fabsf(v3dot(R + 3 * i, R + 3 * j) - (i == j ? 1.f : 0.f)));
  if (bad > 1.0e-3f) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.f : 0.f;
    return true;
  }
  return false;
}
// gR: grad w.r.t. R; gdirect: grad w.r.t. the `unnormalized_6drepr` output (= z itself)
TTK_HD void rot6d_bwd(const float z[6], const float gR[9], const float gdirect[6], float gz[6]) {
  for (int i = 0; i < 6; ++i) gz[i] = gdirect[i];
  float R[9];
  if (rot6d_fwd(z, R)) return;
  const float* x = z;
  const float* y = z + 3;
  float c[3], y2[3], gx[3], gy2[3], gc[3], t[3];
  v3cross(x, y, c);
  v3cross(c, x, y2);
  v3normalize_bwd(x, gR, gx);
  v3normalize_bwd(y2, gR + 3, gy2);
  v3normalize_bwd(c, gR + 6, gc);
  // y2 = c X x: d/dc = x X g, d/dx = g X c
  v3cross(x, gy2, t);
  for (int i = 0; i < 3; ++i) gc[i] += t[i];
  v3cross(gy2, c, t);
  for (int i = 0; i < 3; ++i) gx[i] += t[i];
  // c = x X y: d/dx = y X g, d/dy = g X x
  v3cross(y, gc, t);
  for (int i = 0; i < 3; ++i) gx[i] += t[i];
  v3cross(gc, x, t);
  for (int i = 0; i < 3; ++i) { gz[i] += gx[i]; gz[3 + i] += t[i]; }
}

// LocalToGlobalCoordinateOffset with Mat33Repr (modelcomponents.py:136-184; make_rotate_x: rotrepr.py:73-85, FULL angle)
TTK_HD void offset_fwd_m(const float p[4], const float hR[9], const float hc[3], float R[9], float c[3]) {
  const float sn = sinf(p[1]), cs = cosf(p[1]);
  const float Rx[9] = {1.f, 0.f, 0.f, 0.f, cs, -sn, 0.f, sn, cs};
  const float t[3] = {0.f, p[1], p[2]};
  const float size = hc[2] * elu1(p[3]);
  m3mul(hR, Rx, R);
  float r[3];
  m3vec(hR, t, r);
  c[0] = hc[0] + r[0] * size;
  c[1] = hc[1] + r[1] * size;
  c[2] = size;
}
// accumulates into ghR[9], ghc[3], gp[4]
TTK_HD void offset_bwd_m(const float p[4], const float hR[9], const float hc[3], const float gR[9], const float gc[3],
                         float ghR[9], float ghc[3], float gp[4]) {
  const float sn = sinf(p[1]), cs = cosf(p[1]);
  const float Rx[9] = {1.f, 0.f, 0.f, 0.f, cs, -sn, 0.f, sn, cs};
  const float t[3] = {0.f, p[1], p[2]};
  const float so = elu1(p[3]);
  const float size = hc[2] * so;
  float r[3];
  m3vec(hR, t, r);
  const float gsize = gc[2] + gc[0] * r[0] + gc[1] * r[1];
  const float gr[3] = {gc[0] * size, gc[1] * size, 0.f};
  ghc[0] += gc[0];
  ghc[1] += gc[1];
  ghc[2] += gsize * so;
  gp[3] += gsize * hc[2] * elu1_d(p[3]);
  // R = hR Rx: ghR += gR Rx^T ; gRx = hR^T gR
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float a = 0.f;
      for (int k = 0; k < 3; ++k) a += gR[3 * i + k] * Rx[3 * j + k];
      ghR[3 * i + j] += a;
    }
  float gRx[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) gRx[3 * i + j] = hR[i] * gR[j] + hR[3 + i] * gR[3 + j] + hR[6 + i] * gR[6 + j];
  gp[1] += gRx[4] * (-sn) + gRx[5] * (-cs) + gRx[7] * cs + gRx[8] * (-sn);
  // r = hR t
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) ghR[3 * i + j] += gr[i] * t[j];
  float gt[3];
  m3tvec(hR, gr, gt);
  gp[1] += gt[1];
  gp[2] += gt[2];
}

TTK_HD void landmark_fwd_m(const float Rk[9], const float ck[3], const float local[3], float out[3]) {
  float r[3];
  m3vec(Rk, local, r);
  out[0] = r[0] * ck[2] + ck[0];
  out[1] = r[1] * ck[2] + ck[1];
  out[2] = r[2] * ck[2];
}
// accumulates gRk[9], gck[3]; writes glocal[3]
TTK_HD void landmark_bwd_m(const float Rk[9], const float ck[3], const float local[3], const float g[3], float gRk[9],
                           float gck[3], float glocal[3]) {
  float r[3];
  m3vec(Rk, local, r);
  gck[0] += g[0];
  gck[1] += g[1];
  gck[2] += g[0] * r[0] + g[1] * r[1] + g[2] * r[2];
  const float gr[3] = {g[0] * ck[2], g[1] * ck[2], g[2] * ck[2]};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) gRk[3 * i + j] += gr[i] * local[j];
  m3tvec(Rk, gr, glocal);
}

struct HeadOutM {
  float roi[4], coord[3], Lc[9], Lr[9], ck[3];
  float rot[9], Rk[9];
};
TTK_HD void sample_fwd_core_m(const float* z, bool unc, bool pt, bool use_offset, const float* p, const float* pk,
                              HeadOutM& o) {
  box_fwd(z + Z_BOX, o.roi);
  const float hc[3] = {z[Z_XY], z[Z_XY + 1], elu1(z[Z_SIZE])};
  float hR[9];
  rot6d_fwd(z + Z_QUAT, hR);
  if (unc) {
    tri_scale_fwd(z + z_coord_scale(unc, true), o.Lc);
    tri_scale_fwd(z + z_pose_scale(unc, true), o.Lr);
  }
  if (use_offset) {
    offset_fwd_m(p, hR, hc, o.rot, o.coord);
    if (pt) offset_fwd_m(pk, hR, hc, o.Rk, o.ck);
  } else {
    for (int i = 0; i < 9; ++i) { o.rot[i] = hR[i]; o.Rk[i] = hR[i]; }
    for (int i = 0; i < 3; ++i) { o.coord[i] = hc[i]; o.ck[i] = hc[i]; }
  }
}
struct HeadGradM {
  float roi[4], coord[3], Lc[9], Lr[9], ck[3];
  float rot[9], z6[6], Rk[9];
};
// Writes gz[0 .. z_shape(unc, true)); accumulates gp[4], gpk[4].
TTK_HD void sample_bwd_core_m(const float* z, bool unc, bool pt, bool use_offset, const float* p, const float* pk,
                              const HeadGradM& g, float* gz, float gp[4], float gpk[4]) {
  box_bwd(z + Z_BOX, g.roi, gz + Z_BOX);
  const float hc[3] = {z[Z_XY], z[Z_XY + 1], elu1(z[Z_SIZE])};
  float hR[9];
  rot6d_fwd(z + Z_QUAT, hR);
  float ghR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float ghc[3] = {0.f, 0.f, 0.f};
  if (use_offset) {
    offset_bwd_m(p, hR, hc, g.rot, g.coord, ghR, ghc, gp);
    if (pt) offset_bwd_m(pk, hR, hc, g.Rk, g.ck, ghR, ghc, gpk);
  } else {
    for (int i = 0; i < 9; ++i) ghR[i] = g.rot[i] + g.Rk[i];
    for (int i = 0; i < 3; ++i) ghc[i] = g.coord[i] + g.ck[i];
  }
  gz[Z_XY] = ghc[0];
  gz[Z_XY + 1] = ghc[1];
  gz[Z_SIZE] = ghc[2] * elu1_d(z[Z_SIZE]);
  rot6d_bwd(z + Z_QUAT, ghR, g.z6, gz + Z_QUAT);
  if (unc) {
    tri_scale_bwd(z + z_coord_scale(unc, true), g.Lc, gz + z_coord_scale(unc, true));
    tri_scale_bwd(z + z_pose_scale(unc, true), g.Lr, gz + z_pose_scale(unc, true));
  }
}

}  // namespace hm
}  // namespace ttk
