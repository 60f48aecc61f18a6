"""Checks the device math headers (csrc/head_math.h, csrc/loss_math.h), compiled for the HOST by g++
(tests/host_math/shim.cpp), against autograd of the CPU oracle.  This verifies the hand-derived
backward formulas without a GPU; the product never uses this harness."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import refmodel as R
from oracle.synth import synthetic_keypoint_buffers

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    out = tmp_path_factory.mktemp("hostmath") / "libhostmath.so"
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "host_math", "shim.cpp"), "-o", str(out)])
    return ctypes.CDLL(str(out))


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def f32(*shape, rng, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _identity_state(unc, pt, NZ, rng):
    """State whose linear layers pick rows of z: heads_forward(st, f=z) then acts on z directly."""
    eye = torch.eye(NZ)
    st = {}

    def lin(prefix, lo, n):
        st[prefix + ".weight"] = eye[lo:lo + n].clone()
        st[prefix + ".bias"] = torch.zeros(n)

    lin("boxnet.linear", 0, 4)
    lin("posnet.linear_xy", 4, 2)
    lin("posnet.linear_size", 6, 1)
    lin("quatnet.linear", 7, 4)
    if unc:
        lin("posnet.scales.neck.lin", 11, 7)
        lin("quatnet.uncertainty_net.neck.lin", 18, 7)
        md = torch.tensor([1e-6] * 3 + [0.0] * 3)
        st["posnet.scales.min_diag"] = md
        st["quatnet.uncertainty_net.min_diag"] = md.clone()
        st["boxnet.scales.hidden_scale"] = torch.zeros(5)
        st["landmarks.point_distrib_scales.hidden_scale"] = torch.zeros(69)
        st["landmarks.shape_distrib_scales.hidden_scale"] = torch.zeros(51)
    st["local_pose_offset.p"] = torch.from_numpy(f32(8, 4, rng=rng, scale=0.3)).requires_grad_(True)
    st["local_pose_offset_kpts.p"] = torch.from_numpy(f32(8, 4, rng=rng, scale=0.3)).requires_grad_(True)
    if pt:
        kp, ke = synthetic_keypoint_buffers()
        st["landmarks.deformablekeypoints.keypts"] = torch.from_numpy(kp)
        st["landmarks.deformablekeypoints.keyeigvecs"] = torch.from_numpy(ke)
        lin("landmarks.shapenet", 11 + (14 if unc else 0), 50)
    return st


@pytest.mark.parametrize("unc,pt,use_offset", [(True, True, True), (False, True, True), (False, False, True), (True, True, False)])
def test_heads_forward_backward(hm, unc, pt, use_offset):
    rng = np.random.default_rng(3)
    n = 6
    NZ = 11 + (14 if unc else 0) + (50 if pt else 0)
    z = f32(n, NZ, rng=rng, scale=0.7)
    ids = rng.integers(0, 8, n).astype(np.int32)
    st = _identity_state(unc, pt, NZ, rng)
    zt = torch.from_numpy(z).requires_grad_(True)
    out = R.heads_forward(st, zt, torch.from_numpy(ids), enable_point_head=pt, enable_uncertainty=unc,
                          use_local_pose_offset=use_offset, training=True)
    names = ["roi", "coord", "rot", "unnormalized_quat"] + (["coord_scales", "pose_scales_tril"] if unc else []) + (["pt3d_68", "shapeparam"] if pt else [])
    ups = {k: torch.from_numpy(f32(*out[k].shape, rng=rng)) for k in names}
    sum((out[k] * ups[k]).sum() for k in names).backward()

    kp, ke = synthetic_keypoint_buffers()
    Pm, Pk = st["local_pose_offset.p"].detach().numpy().copy(), st["local_pose_offset_kpts.p"].detach().numpy().copy()
    o = {"roi": np.zeros((n, 4), np.float32), "coord": np.zeros((n, 3), np.float32), "rot": np.zeros((n, 4), np.float32),
         "unnormalized_quat": np.zeros((n, 4), np.float32), "coord_scales": np.zeros((n, 9), np.float32),
         "pose_scales_tril": np.zeros((n, 9), np.float32), "pt3d_68": np.zeros((n, 68, 3), np.float32)}
    hm.hm_heads_fwd(n, NZ, P(z), P(ids), P(Pm), P(Pk), P(kp), P(ke), int(unc), int(pt), int(use_offset), P(o["roi"]), P(o["coord"]),
                    P(o["rot"]), P(o["unnormalized_quat"]), P(o["coord_scales"]), P(o["pose_scales_tril"]), P(o["pt3d_68"]))
    for k in names:
        if k == "shapeparam":
            continue
        np.testing.assert_allclose(o[k].reshape(out[k].shape), out[k].detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)

    def up(k, shape):
        return np.ascontiguousarray(ups[k].numpy().reshape(shape)) if k in ups else np.zeros(shape, np.float32)

    gz, gP, gPk = np.zeros((n, NZ), np.float32), np.zeros((8, 4), np.float32), np.zeros((8, 4), np.float32)
    hm.hm_heads_bwd(n, NZ, P(z), P(ids), P(Pm), P(Pk), P(kp), P(ke), int(unc), int(pt), int(use_offset), P(up("roi", (n, 4))),
                    P(up("coord", (n, 3))), P(up("rot", (n, 4))), P(up("unnormalized_quat", (n, 4))), P(up("coord_scales", (n, 9))),
                    P(up("pose_scales_tril", (n, 9))), P(up("pt3d_68", (n, 68, 3))), P(up("shapeparam", (n, 50))), P(gz), P(gP), P(gPk))
    np.testing.assert_allclose(gz, zt.grad.numpy(), rtol=2e-4, atol=2e-5)
    if use_offset:
        np.testing.assert_allclose(gP, st["local_pose_offset.p"].grad.numpy(), rtol=2e-4, atol=2e-5)
        if pt:
            np.testing.assert_allclose(gPk, st["local_pose_offset_kpts.p"].grad.numpy(), rtol=2e-4, atol=2e-5)


def _unit(a):
    return a / np.linalg.norm(a, axis=-1, keepdims=True)


def _tril(rng, n):
    L = np.zeros((n, 3, 3), np.float32)
    L[:, [0, 1, 2], [0, 1, 2]] = rng.uniform(0.3, 1.5, (n, 3))
    L[:, 1, 0], L[:, 2, 0], L[:, 2, 1] = rng.normal(0, 0.3, (3, n))
    return L


def test_rotation_losses(hm):
    rng = np.random.default_rng(11)
    n = 64
    q, t = _unit(f32(n, 4, rng=rng)), _unit(f32(n, 4, rng=rng))
    t[:8] = q[:8] + 1e-3 * f32(8, 4, rng=rng)  # nearly identical rotations (small angle branch)
    t = _unit(t).astype(np.float32)
    L = _tril(rng, n)
    gv = f32(n, rng=rng)
    for name, ref in (("lm_rot", R.loss_rot), ("lm_nllrot", R.loss_nllrot)):
        qt, Lt = torch.from_numpy(q).requires_grad_(True), torch.from_numpy(L).requires_grad_(True)
        v_ref = ref({"rot": qt, "pose_scales_tril": Lt}, {"pose": torch.from_numpy(t)})
        (v_ref * torch.from_numpy(gv)).sum().backward()
        v, gq, gL = np.zeros(n, np.float32), np.zeros((n, 4), np.float32), np.zeros((n, 3, 3), np.float32)
        if name == "lm_rot":
            hm.lm_rot(n, P(q), P(t), P(gv), P(v), P(gq))
        else:
            hm.lm_nllrot(n, P(q), P(t), P(L), P(gv), P(v), P(gq), P(gL))
            np.testing.assert_allclose(gL, Lt.grad.numpy(), rtol=3e-4, atol=3e-5)
        np.testing.assert_allclose(v, v_ref.detach().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(gq, qt.grad.numpy(), rtol=3e-4, atol=3e-5)


def test_coord_nll_and_quatreg_and_normal(hm):
    rng = np.random.default_rng(12)
    n = 64
    c, t, L, gv = f32(n, 3, rng=rng), f32(n, 3, rng=rng), _tril(rng, n), f32(n, rng=rng)
    ct, Lt = torch.from_numpy(c).requires_grad_(True), torch.from_numpy(L).requires_grad_(True)
    v_ref = R.loss_nllcoord({"coord": ct, "coord_scales": Lt}, {"coord": torch.from_numpy(t)})
    (v_ref * torch.from_numpy(gv)).sum().backward()
    v, gc, gL = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3, 3), np.float32)
    hm.lm_nllcoord(n, P(c), P(t), P(L), P(gv), P(v), P(gc), P(gL))
    np.testing.assert_allclose(v, v_ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(gc, ct.grad.numpy(), rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(gL, Lt.grad.numpy(), rtol=3e-4, atol=3e-5)

    qu = f32(n, 4, rng=rng)
    qt = torch.from_numpy(qu).requires_grad_(True)
    v_ref = R.loss_quatreg({"unnormalized_quat": qt}, None)
    (v_ref * torch.from_numpy(gv)).sum().backward()
    gq = np.zeros((n, 4), np.float32)
    hm.lm_quatreg(n, P(qu), P(gv), P(v), P(gq))
    np.testing.assert_allclose(v, v_ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(gq, qt.grad.numpy(), rtol=3e-4, atol=3e-5)

    mu, sg, x = f32(n, rng=rng), rng.uniform(0.2, 2.0, n).astype(np.float32), f32(n, rng=rng)
    mt, st_ = torch.from_numpy(mu).requires_grad_(True), torch.from_numpy(sg).requires_grad_(True)
    v_ref = -R._normal_logprob(torch.from_numpy(x), mt, st_)
    v_ref.sum().backward()
    gmu, gsg = np.zeros(n, np.float32), np.zeros(n, np.float32)
    hm.lm_normal(n, P(mu), P(sg), P(x), P(v), P(gmu), P(gsg))
    np.testing.assert_allclose(v, v_ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(gmu, mt.grad.numpy(), rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(gsg, st_.grad.numpy(), rtol=3e-4, atol=3e-5)


def test_gmm_and_point_weights(hm, golden_dir):
    rng = np.random.default_rng(13)
    n = 16
    gmm = R.ShapeGmm(os.path.join(golden_dir, "shapeparams_gmm.npz"))
    x = f32(n, 50, rng=rng, scale=0.5)
    xt = torch.from_numpy(x).requires_grad_(True)
    v_ref = gmm({"shapeparam": xt}, None)
    v_ref.sum().backward()
    K = gmm.w.shape[0]
    ck = (torch.log(gmm.w) + torch.log(gmm.sinv).sum(-1) - gmm.normc).numpy().copy()
    mu, sinv = gmm.mu.numpy().copy(), gmm.sinv.numpy().copy()
    v, post = np.zeros(n, np.float32), np.zeros((n, K), np.float64)
    hm.lm_gmm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    hm.lm_gmm(n, P(x), P(ck), P(mu), P(sinv), K, gmm.fudge, P(v), P(post))
    np.testing.assert_allclose(v, v_ref.detach().numpy(), rtol=1e-6, atol=1e-7)
    # gradient: fudge * sum_k post_k (x - mu_k) sinv_k^2
    g = gmm.fudge * np.einsum("nk,nkd->nd", post, (x[:, None, :].astype(np.float64) - mu[None]) * sinv[None] ** 2)
    np.testing.assert_allclose(g, xt.grad.numpy(), rtol=1e-4, atol=1e-8)
    w = np.zeros(68, np.float32)
    hm.lm_point_weights.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    hm.lm_point_weights(0.8, 0.0, P(w))
    np.testing.assert_array_equal(w, R.point_weights(0.8, 0.0).numpy())


# ---------------------------------------------------------------------------------------------
# 6D rotation head (RotRepr6dWithNormalization) and its losses
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("unc,pt,use_offset", [(True, True, True), (False, True, True), (False, False, True), (True, True, False)])
def test_heads6d_forward_backward(hm, unc, pt, use_offset):
    rng = np.random.default_rng(5)
    n = 6
    NZ = 13 + (14 if unc else 0) + (50 if pt else 0)
    z = f32(n, NZ, rng=rng, scale=0.7)
    z[1, 10:13] = z[1, 7:10] * 1.5  # a collinear (x, y) pair: degenerate Gram-Schmidt -> identity fallback, no gradient
    ids = rng.integers(0, 8, n).astype(np.int32)
    st = _identity_state(unc, pt, NZ, rng)
    eye = torch.eye(NZ)
    # re-map the row pickers to the 6D layout (6 rotation rows, everything after shifted by 2)
    st["quatnet.linear.weight"], st["quatnet.linear.bias"] = eye[7:13].clone(), torch.zeros(6)
    if unc:
        st["posnet.scales.neck.lin.weight"], st["quatnet.uncertainty_net.neck.lin.weight"] = eye[13:20].clone(), eye[20:27].clone()
    if pt:
        lo = 13 + (14 if unc else 0)
        st["landmarks.shapenet.weight"] = eye[lo:lo + 50].clone()
    zt = torch.from_numpy(z).requires_grad_(True)
    out = R.heads_forward(st, zt, torch.from_numpy(ids), enable_point_head=pt, enable_uncertainty=unc,
                          use_local_pose_offset=use_offset, training=True, enable_6drot=True)
    names = ["roi", "coord", "rot", "unnormalized_6drepr"] + (["coord_scales", "pose_scales_tril"] if unc else []) + (["pt3d_68", "shapeparam"] if pt else [])
    ups = {k: torch.from_numpy(f32(*out[k].shape, rng=rng)) for k in names}
    sum((out[k] * ups[k]).sum() for k in names).backward()

    kp, ke = synthetic_keypoint_buffers()
    Pm, Pk = st["local_pose_offset.p"].detach().numpy().copy(), st["local_pose_offset_kpts.p"].detach().numpy().copy()
    o = {"roi": np.zeros((n, 4), np.float32), "coord": np.zeros((n, 3), np.float32), "rot": np.zeros((n, 9), np.float32),
         "coord_scales": np.zeros((n, 9), np.float32), "pose_scales_tril": np.zeros((n, 9), np.float32),
         "pt3d_68": np.zeros((n, 68, 3), np.float32)}
    hm.hm_heads6d_fwd(n, NZ, P(z), P(ids), P(Pm), P(Pk), P(kp), P(ke), int(unc), int(pt), int(use_offset), P(o["roi"]), P(o["coord"]),
                      P(o["rot"]), P(o["coord_scales"]), P(o["pose_scales_tril"]), P(o["pt3d_68"]))
    for k in names:
        if k in ("shapeparam", "unnormalized_6drepr"):
            continue
        np.testing.assert_allclose(o[k].reshape(out[k].shape), out[k].detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    if not use_offset:
        np.testing.assert_array_equal(o["rot"][1].reshape(3, 3), np.eye(3, dtype=np.float32))  # the fallback sample

    def up(k, shape):
        return np.ascontiguousarray(ups[k].numpy().reshape(shape)) if k in ups else np.zeros(shape, np.float32)

    gz, gP, gPk = np.zeros((n, NZ), np.float32), np.zeros((8, 4), np.float32), np.zeros((8, 4), np.float32)
    hm.hm_heads6d_bwd(n, NZ, P(z), P(ids), P(Pm), P(Pk), P(kp), P(ke), int(unc), int(pt), int(use_offset), P(up("roi", (n, 4))),
                      P(up("coord", (n, 3))), P(up("rot", (n, 9))), P(up("unnormalized_6drepr", (n, 6))), P(up("coord_scales", (n, 9))),
                      P(up("pose_scales_tril", (n, 9))), P(up("pt3d_68", (n, 68, 3))), P(up("shapeparam", (n, 50))), P(gz), P(gP), P(gPk))
    np.testing.assert_allclose(gz, zt.grad.numpy(), rtol=2e-4, atol=3e-5)
    if use_offset:
        np.testing.assert_allclose(gP, st["local_pose_offset.p"].grad.numpy(), rtol=2e-4, atol=3e-5)
        if pt:
            np.testing.assert_allclose(gPk, st["local_pose_offset_kpts.p"].grad.numpy(), rtol=2e-4, atol=3e-5)


def test_rot6d_losses_and_from_matrix(hm):
    rng = np.random.default_rng(13)
    n = 256
    z6 = f32(n, 6, rng=rng)
    Rm = R.rot6d_to_matrix(torch.from_numpy(z6)).numpy().copy()
    tq = _unit(f32(n, 4, rng=rng)).astype(np.float32)
    gv = f32(n, rng=rng)
    # Rot6dReprLoss
    Rt = torch.from_numpy(Rm).requires_grad_(True)
    v_ref = R.loss_rot6d({"rot": Rt}, {"pose": torch.from_numpy(tq)})
    (v_ref * torch.from_numpy(gv)).sum().backward()
    v, gR = np.zeros(n, np.float32), np.zeros((n, 9), np.float32)
    hm.lm_rot6d(n, P(Rm), P(tq), P(gv), P(v), P(gR))
    np.testing.assert_allclose(v, v_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gR.reshape(n, 3, 3), Rt.grad.numpy(), rtol=1e-5, atol=1e-6)
    # Rot6dNormalizationSoftConstraint
    zt = torch.from_numpy(z6).requires_grad_(True)
    v_ref = R.loss_ortho6d({"unnormalized_6drepr": zt}, None)
    (v_ref * torch.from_numpy(gv)).sum().backward()
    gz = np.zeros((n, 6), np.float32)
    hm.lm_ortho6d(n, P(z6), P(gv), P(v), P(gz))
    np.testing.assert_allclose(v, v_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gz, zt.grad.numpy(), rtol=1e-5, atol=1e-6)
    # Mat33Repr.as_quat = torchquaternion.from_matrix: all four branches occur in 256 random rotations
    Rt = torch.from_numpy(Rm).requires_grad_(True)
    q_ref = R.matrix_to_quat(Rt)
    gq = f32(n, 4, rng=rng)
    (q_ref * torch.from_numpy(gq)).sum().backward()
    q, gm = np.zeros((n, 4), np.float32), np.zeros((n, 9), np.float32)
    hm.lm_from_matrix(n, P(Rm), P(gq), P(q), P(gm))
    d0, d1, d2 = Rm[:, 0, 0], Rm[:, 1, 1], Rm[:, 2, 2]
    picks = np.argmax(np.stack([-d0 - d1 + d2, -d0 + d1 - d2, d0 - d1 - d2, d0 + d1 + d2], -1), -1)
    assert set(picks.tolist()) == {0, 1, 2, 3}
    np.testing.assert_allclose(q, q_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gm.reshape(n, 3, 3), Rt.grad.numpy(), rtol=2e-4, atol=2e-5)
    # and it inverts tomatrix
    np.testing.assert_allclose(np.abs((q * tq).sum(-1)) <= 1.0 + 1e-5, True)
