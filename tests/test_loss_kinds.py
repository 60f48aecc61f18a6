"""The non-default kinds of the reference's loss switches (losses.py:16-39 "l1" / "smooth_l1" / "smooth_geodesic", negloglikelihood.py:68-69
"laplace") against tests/golden/loss_kinds.npz - values and gradients produced by the reference's own loss classes
(oracle/tools/gen_golden_losses.py).  CPU: the oracle restatement and the device math header compiled for the host; GPU
(tests/test_loss_kinds_gpu.py): the product's loss classes through the C-ABI."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import refmodel as R

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "loss_kinds.npz"))
T = lambda k: torch.from_numpy(G["in/" + k].copy())

# case name -> (oracle function(pred, sample), prediction tensors by key, sample)
def cases():
    c = {}
    for kind in ("l1", "smooth_l1"):
        c[f"xy/{kind}"] = (lambda p, s, k=kind: R.loss_xy_kind(p, s, k), {"coord": "coord_p"}, {"coord": "coord_t"})
        c[f"size/{kind}"] = (lambda p, s, k=kind: R.loss_sz_kind(p, s, k), {"coord": "coord_p"}, {"coord": "coord_t"})
        c[f"box/{kind}"] = (lambda p, s, k=kind: R.loss_box_kind(p, s, k), {"roi": "roi_p"}, {"roi": "roi_t"})
        for dim in (2, 3):
            c[f"points{dim}/{kind}"] = (lambda p, s, k=kind, d=dim: R.loss_points3d_kind(p, s, k, d), {"pt3d_68": "pts_p"}, {"pt3d_68": "pts_t"})
    c["rot/smooth_geodesic"] = (R.loss_rot_smooth_geodesic, {"rot": "quat_p"}, {"pose": "quat_t"})
    for dist in ("laplace", "gaussian"):
        c[f"nllcoord_indep/{dist}"] = (lambda p, s, d=dist: R.loss_nllcoord_indep(p, s, 1.0, 0.5, d), {"coord": "coord_p", "coord_scales": "coord_s"}, {"coord": "coord_t"})
    c["nllbox/laplace"] = (lambda p, s: R.loss_nllbox_dist(p, s, "laplace"), {"roi": "roi_p", "roi_scales": "roi_s"}, {"roi": "roi_t"})
    for dim in (2, 3):
        c[f"nllpoints{dim}/laplace"] = (lambda p, s, d=dim: R.loss_nllpoints3d_dist(p, s, "laplace", d), {"pt3d_68": "pts_p", "pt3d_68_scales": "pts_s"}, {"pt3d_68": "pts_t"})
    c["nllshape/laplace"] = (lambda p, s: R.loss_nllshape_dist(p, s, "laplace"), {"shapeparam": "shape_p", "shapeparam_scales": "shape_s"}, {"shapeparam": "shape_t"})
    return c


CASES = cases()


def test_golden_covers_every_case():
    names = {k.rsplit("/values", 1)[0] for k in G.files if k.endswith("/values")}
    assert names == set(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_golden(name):
    fn, pk, sk = CASES[name]
    pred = {k: T(v).requires_grad_(True) for k, v in pk.items()}
    vals = fn(pred, {k: T(v) for k, v in sk.items()})
    (vals * T("cot")).sum().backward()
    np.testing.assert_allclose(vals.detach().numpy(), G[name + "/values"], rtol=1e-5, atol=1e-7)
    for k, t in pred.items():
        np.testing.assert_allclose(t.grad.numpy(), G[f"{name}/grad/{k}"], rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    out = tmp_path_factory.mktemp("hostmath_kinds") / "libhostmath.so"
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "host_math", "shim.cpp"), "-o", str(out)])
    return ctypes.CDLL(str(out))


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_device_math_header_on_host(hm):
    """csrc/loss_math.h's elem_loss / laplace_nll / smooth_geodesic_loss and their hand-derived derivatives, compiled by g++."""
    n = G["in/cot"].shape[0]
    # elementwise kinds on the box tensors: values summed with the 1/4 mean weights, derivative per element
    p, t = G["in/roi_p"].reshape(-1).copy(), G["in/roi_t"].reshape(-1).copy()
    for kind, code in (("l1", 1), ("smooth_l1", 2)):
        v, d = np.zeros_like(p), np.zeros_like(p)
        hm.lm_elem(p.size, code, ctypes.c_float(0.01), P(p), P(t), P(v), P(d))
        np.testing.assert_allclose(v.reshape(n, 4).mean(-1), G[f"box/{kind}/values"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(d.reshape(n, 4) * G["in/cot"][:, None] / 4.0, G[f"box/{kind}/grad/roi"], rtol=1e-5, atol=1e-7)
    # Laplace on the shape parameters
    mu, b, x = (G["in/" + k].reshape(-1).copy() for k in ("shape_p", "shape_s", "shape_t"))
    v, gmu, gb = np.zeros_like(mu), np.zeros_like(mu), np.zeros_like(mu)
    hm.lm_laplace(mu.size, P(mu), P(b), P(x), P(v), P(gmu), P(gb))
    np.testing.assert_allclose(v.reshape(n, 50).mean(-1), G["nllshape/laplace/values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gmu.reshape(n, 50) * G["in/cot"][:, None] / 50.0, G["nllshape/laplace/grad/shapeparam"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gb.reshape(n, 50) * G["in/cot"][:, None] / 50.0, G["nllshape/laplace/grad/shapeparam_scales"], rtol=2e-5, atol=1e-6)
    # smooth geodesic distance
    q, tq, gv = G["in/quat_p"].copy(), G["in/quat_t"].copy(), G["in/cot"].copy()
    v, gq = np.zeros(n, np.float32), np.zeros((n, 4), np.float32)
    hm.lm_rot_geodesic(n, P(q), P(tq), P(gv), P(v), P(gq))
    np.testing.assert_allclose(v, G["rot/smooth_geodesic/values"], rtol=2e-4, atol=2e-7)
    np.testing.assert_allclose(gq, G["rot/smooth_geodesic/grad/rot"], rtol=2e-3, atol=2e-5)
