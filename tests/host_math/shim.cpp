// TEST-ONLY host harness: compiles the device math headers (csrc/head_math.h, csrc/loss_math.h) with
// g++ and exposes per-sample loops through a C interface so that tests/test_host_math.py can check
// every hand-derived gradient against autograd of the CPU oracle WITHOUT a GPU.  Never linked into
// libttk_hip.so and never used by the product.
#include <string.h>

#include "../../neuralnet-tracker-traincode_amd/csrc/loss_math.h"

using namespace ttk;
using hm::Q;

extern "C" {

// outputs: roi[n,4] coord[n,3] rot[n,4] qu[n,4] Lc[n,9] Lr[n,9] pts[n,68,3]
void hm_heads_fwd(int n, int NZ, const float* z, const int* ids, const float* P, const float* Pk, const float* kp,
                  const float* eig, int unc, int pt, int use_offset, float* roi, float* coord, float* rot, float* qu,
                  float* Lc, float* Lr, float* pts) {
  for (int s = 0; s < n; ++s) {
    const float* zs = z + (size_t)s * NZ;
    hm::HeadOut o;
    const int id = ids ? ids[s] : 0;
    hm::sample_fwd_core(zs, unc, pt, use_offset, P + 4 * id, Pk + 4 * id, o);
    memcpy(roi + 4 * s, o.roi, 16);
    memcpy(coord + 3 * s, o.coord, 12);
    memcpy(rot + 4 * s, &o.rot, 16);
    memcpy(qu + 4 * s, &o.qu, 16);
    if (unc) { memcpy(Lc + 9 * s, o.Lc, 36); memcpy(Lr + 9 * s, o.Lr, 36); }
    if (pt) {
      const float* shp = zs + hm::z_shape(unc);
      for (int p = 0; p < 68; ++p) {
        float local[3];
        for (int d = 0; d < 3; ++d) {
          float a = kp[p * 3 + d];
          for (int i = 0; i < 50; ++i) a += eig[(i * 68 + p) * 3 + d] * shp[i];
          local[d] = a;
        }
        hm::landmark_fwd(o.qk, o.ck, local, pts + ((size_t)s * 68 + p) * 3);
      }
    }
  }
}

// upstream grads in the same layout (+ gshp_out[n,50]); outputs gz[n,NZ], gP[8,4], gPk[8,4]
void hm_heads_bwd(int n, int NZ, const float* z, const int* ids, const float* P, const float* Pk, const float* kp,
                  const float* eig, int unc, int pt, int use_offset, const float* g_roi, const float* g_coord,
                  const float* g_rot, const float* g_qu, const float* g_Lc, const float* g_Lr, const float* g_pts,
                  const float* g_shp, float* gz, float* gP, float* gPk) {
  memset(gP, 0, 32 * sizeof(float));
  memset(gPk, 0, 32 * sizeof(float));
  for (int s = 0; s < n; ++s) {
    const float* zs = z + (size_t)s * NZ;
    float* gzs = gz + (size_t)s * NZ;
    const int id = ids ? ids[s] : 0;
    hm::HeadOut o;
    hm::sample_fwd_core(zs, unc, pt, use_offset, P + 4 * id, Pk + 4 * id, o);
    hm::HeadGrad g;
    memset(&g, 0, sizeof(g));
    memcpy(g.roi, g_roi + 4 * s, 16);
    memcpy(g.coord, g_coord + 3 * s, 12);
    memcpy(&g.rot, g_rot + 4 * s, 16);
    memcpy(&g.qu, g_qu + 4 * s, 16);
    if (unc) { memcpy(g.Lc, g_Lc + 9 * s, 36); memcpy(g.Lr, g_Lr + 9 * s, 36); }
    if (pt) {
      const float* shp = zs + hm::z_shape(unc);
      float* gshp = gzs + hm::z_shape(unc);
      for (int i = 0; i < 50; ++i) gshp[i] = g_shp[s * 50 + i];
      for (int p = 0; p < 68; ++p) {
        float local[3], gl[3];
        for (int d = 0; d < 3; ++d) {
          float a = kp[p * 3 + d];
          for (int i = 0; i < 50; ++i) a += eig[(i * 68 + p) * 3 + d] * shp[i];
          local[d] = a;
        }
        hm::landmark_bwd(o.qk, o.ck, local, g_pts + ((size_t)s * 68 + p) * 3, g.qk, g.ck, gl);
        for (int i = 0; i < 50; ++i)
          for (int d = 0; d < 3; ++d) gshp[i] += gl[d] * eig[(i * 68 + p) * 3 + d];
      }
    }
    hm::sample_bwd_core(zs, unc, pt, use_offset, P + 4 * id, Pk + 4 * id, g, gzs, gP + 4 * id, gPk + 4 * id);
  }
}

void lm_rot(int n, const float* q, const float* t, const float* gv, float* v, float* gq) {
  for (int s = 0; s < n; ++s) { v[s] = lm::rot_loss(q + 4 * s, t + 4 * s); lm::rot_loss_bwd(q + 4 * s, t + 4 * s, gv[s], gq + 4 * s); }
}
void lm_quatreg(int n, const float* q, const float* gv, float* v, float* gq) {
  for (int s = 0; s < n; ++s) { v[s] = lm::quatreg_loss(q + 4 * s); lm::quatreg_loss_bwd(q + 4 * s, gv[s], gq + 4 * s); }
}
void lm_nllrot(int n, const float* q, const float* t, const float* L, const float* gv, float* v, float* gq, float* gL) {
  for (int s = 0; s < n; ++s) {
    v[s] = lm::nllrot_loss(q + 4 * s, t + 4 * s, L + 9 * s);
    lm::nllrot_loss_bwd(q + 4 * s, t + 4 * s, L + 9 * s, gv[s], gq + 4 * s, gL + 9 * s);
  }
}
void lm_nllcoord(int n, const float* c, const float* t, const float* L, const float* gv, float* v, float* gc, float* gL) {
  for (int s = 0; s < n; ++s) {
    v[s] = lm::nllcoord_loss(c + 3 * s, t + 3 * s, L + 9 * s);
    lm::nllcoord_loss_bwd(c + 3 * s, t + 3 * s, L + 9 * s, gv[s], gc + 3 * s, gL + 9 * s);
  }
}
void lm_normal(int n, const float* mu, const float* sg, const float* x, float* v, float* gmu, float* gsg) {
  for (int s = 0; s < n; ++s) { v[s] = lm::normal_nll(mu[s], sg[s], x[s]); lm::normal_nll_bwd(mu[s], sg[s], x[s], 1.f, gmu[s], gsg[s]); }
}
// the non-default loss kinds: elementwise distance kinds, Laplace NLL, smooth geodesic distance
void lm_elem(int n, int kind, float beta, const float* p, const float* t, float* v, float* d) {
  for (int s = 0; s < n; ++s) { v[s] = lm::elem_loss(kind, p[s] - t[s], beta); d[s] = lm::elem_loss_d(kind, p[s] - t[s], beta); }
}
void lm_laplace(int n, const float* mu, const float* b, const float* x, float* v, float* gmu, float* gb) {
  for (int s = 0; s < n; ++s) { v[s] = lm::laplace_nll(mu[s], b[s], x[s]); lm::laplace_nll_bwd(mu[s], b[s], x[s], 1.f, gmu[s], gb[s]); }
}
void lm_rot_geodesic(int n, const float* q, const float* t, const float* gv, float* v, float* gq) {
  for (int s = 0; s < n; ++s) { v[s] = lm::smooth_geodesic_loss(q + 4 * s, t + 4 * s); lm::smooth_geodesic_loss_bwd(q + 4 * s, t + 4 * s, gv[s], gq + 4 * s); }
}
void lm_gmm(int n, const float* x, const double* ck, const double* mu, const double* sinv, int K, double fudge, float* v,
            double* post) {
  for (int s = 0; s < n; ++s) v[s] = (float)lm::gmm_nll(x + 50 * s, ck, mu, sinv, K, fudge, post + K * s);
}
void lm_point_weights(float chin, float eye, float* w) {
  for (int p = 0; p < 68; ++p) w[p] = lm::point_weight(p, chin, eye);
}
// ---- 6D rotation head (matrix mode).  outputs: roi[n,4] coord[n,3] rot[n,9] Lc[n,9] Lr[n,9] pts[n,68,3]
void hm_heads6d_fwd(int n, int NZ, const float* z, const int* ids, const float* P, const float* Pk, const float* kp,
                    const float* eig, int unc, int pt, int use_offset, float* roi, float* coord, float* rot, float* Lc,
                    float* Lr, float* pts) {
  for (int s = 0; s < n; ++s) {
    const float* zs = z + (size_t)s * NZ;
    hm::HeadOutM o;
    const int id = ids ? ids[s] : 0;
    hm::sample_fwd_core_m(zs, unc, pt, use_offset, P + 4 * id, Pk + 4 * id, o);
    memcpy(roi + 4 * s, o.roi, 16);
    memcpy(coord + 3 * s, o.coord, 12);
    memcpy(rot + 9 * s, o.rot, 36);
    if (unc) { memcpy(Lc + 9 * s, o.Lc, 36); memcpy(Lr + 9 * s, o.Lr, 36); }
    if (pt) {
      const float* shp = zs + hm::z_shape(unc, true);
      for (int p = 0; p < 68; ++p) {
        float local[3];
        for (int d = 0; d < 3; ++d) {
          float a = kp[p * 3 + d];
          for (int i = 0; i < 50; ++i) a += eig[(i * 68 + p) * 3 + d] * shp[i];
          local[d] = a;
        }
        hm::landmark_fwd_m(o.Rk, o.ck, local, pts + ((size_t)s * 68 + p) * 3);
      }
    }
  }
}
void hm_heads6d_bwd(int n, int NZ, const float* z, const int* ids, const float* P, const float* Pk, const float* kp,
                    const float* eig, int unc, int pt, int use_offset, const float* g_roi, const float* g_coord,
                    const float* g_rot, const float* g_z6, const float* g_Lc, const float* g_Lr, const float* g_pts,
                    const float* g_shp, float* gz, float* gP, float* gPk) {
  memset(gP, 0, 32 * sizeof(float));
  memset(gPk, 0, 32 * sizeof(float));
  for (int s = 0; s < n; ++s) {
    const float* zs = z + (size_t)s * NZ;
    float* gzs = gz + (size_t)s * NZ;
    const int id = ids ? ids[s] : 0;
    hm::HeadOutM o;
    hm::sample_fwd_core_m(zs, unc, pt, use_offset, P + 4 * id, Pk + 4 * id, o);
    hm::HeadGradM g;
    memset(&g, 0, sizeof(g));
    memcpy(g.roi, g_roi + 4 * s, 16);
    memcpy(g.coord, g_coord + 3 * s, 12);
    memcpy(g.rot, g_rot + 9 * s, 36);
    memcpy(g.z6, g_z6 + 6 * s, 24);
    if (unc) { memcpy(g.Lc, g_Lc + 9 * s, 36); memcpy(g.Lr, g_Lr + 9 * s, 36); }
    if (pt) {
      const float* shp = zs + hm::z_shape(unc, true);
      float* gshp = gzs + hm::z_shape(unc, true);
      for (int i = 0; i < 50; ++i) gshp[i] = g_shp[s * 50 + i];
      for (int p = 0; p < 68; ++p) {
        float local[3], gl[3];
        for (int d = 0; d < 3; ++d) {
          float a = kp[p * 3 + d];
          for (int i = 0; i < 50; ++i) a += eig[(i * 68 + p) * 3 + d] * shp[i];
          local[d] = a;
        }
        hm::landmark_bwd_m(o.Rk, o.ck, local, g_pts + ((size_t)s * 68 + p) * 3, g.Rk, g.ck, gl);
        for (int i = 0; i < 50; ++i)
          for (int d = 0; d < 3; ++d) gshp[i] += gl[d] * eig[(i * 68 + p) * 3 + d];
      }
    }
    hm::sample_bwd_core_m(zs, unc, pt, use_offset, P + 4 * id, Pk + 4 * id, g, gzs, gP + 4 * id, gPk + 4 * id);
  }
}
void lm_rot6d(int n, const float* R, const float* t, const float* gv, float* v, float* gR) {
  for (int s = 0; s < n; ++s) { v[s] = lm::rot6d_loss(R + 9 * s, t + 4 * s); lm::rot6d_loss_bwd(t + 4 * s, gv[s], gR + 9 * s); }
}
void lm_ortho6d(int n, const float* z, const float* gv, float* v, float* gz) {
  for (int s = 0; s < n; ++s) { v[s] = lm::ortho6d_loss(z + 6 * s); lm::ortho6d_loss_bwd(z + 6 * s, gv[s], gz + 6 * s); }
}
void lm_from_matrix(int n, const float* m, const float* gq, float* q, float* gm) {
  for (int s = 0; s < n; ++s) { lm::from_matrix(m + 9 * s, q + 4 * s); lm::from_matrix_bwd(m + 9 * s, gq + 4 * s, gm + 9 * s); }
}
}
