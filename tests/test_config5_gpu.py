"""BASELINE config 5, fp32 leg, at full size on one GPU: 512 crops per step drawn from four HBM-resident uint8 datasets of
192x192 frames with the reference's sampling weights 60 000 : 40 000 : 10 000 : 10 000 (pipelines.py:399-453), random
view ROIs / angles (batch/geometric.py:63-84), GPU warp to 129x129 + label bookkeeping - every sample checked against the
CPU oracle (oracle/augment.py, pinned to the reference by tests/golden/augment.npz): view_roi BIT-EXACT (integer
bookkeeping), transforms, all labels, and the warped pixels."""
import numpy as np
import pytest
import torch

from oracle import augment as A

pytestmark = pytest.mark.gpu
DEV = "cuda"
N = 129


def _dataset(tag, n, seed, with_shape):
    g = np.random.default_rng(seed)
    img = g.integers(0, 256, (n, 1, 192, 192), dtype=np.uint8)
    yy, xx = np.mgrid[0:192, 0:192]
    img = (0.5 * img + 0.5 * (127 + 100 * np.sin(xx[None, None] * g.uniform(0.02, 0.2, (n, 1, 1, 1)) + yy[None, None] * g.uniform(0.02, 0.2, (n, 1, 1, 1))))).astype(np.uint8)
    c = g.uniform(60, 132, (n, 2))
    half = g.uniform(25, 60, (n, 2))
    roi = np.concatenate([c - half, c + half], -1).astype(np.float32)  # face boxes in pixels
    pose = g.standard_normal((n, 4)).astype(np.float32)
    pose /= np.linalg.norm(pose, axis=-1, keepdims=True)
    coord = np.concatenate([c, half.mean(-1, keepdims=True)], -1).astype(np.float32)
    pts = np.concatenate([c[:, None, :] + g.standard_normal((n, 68, 2)) * half[:, None, :] * 0.5, g.standard_normal((n, 68, 1)) * 20], -1).astype(np.float32)
    f = {"image": img, "roi": roi, "pose": pose, "coord": coord, "pt3d_68": pts, "frame_id": np.arange(n, dtype=np.int64) + 1000000 * seed}
    if with_shape:
        f["shapeparam"] = (g.standard_normal((n, 50)) * 0.5).astype(np.float32)
    return tag, f


class _RecordingParams:
    """MakeRoiRandomizationParameters that keeps what it drew, in call order (one call per output sub-batch)."""

    def __init__(self):
        from trackertraincode.datatransformation.batch.geometric import MakeRoiRandomizationParameters

        self.inner, self.calls = MakeRoiRandomizationParameters(30.0, 1.1), []

    def __call__(self, B, generator=None, device="cpu"):
        p = self.inner(B, generator=generator, device=device)
        self.calls.append(p)
        return p


def test_multitask_mix_through_resident_loader_matches_oracle_per_sample():
    from trackertraincode.datasets.resident import ResidentFrames, ResidentLoader
    from trackertraincode.datatransformation import GpuFocusRoiAugment
    from trackertraincode.pipelines import Tag

    specs = [_dataset(Tag.POSE_WITH_LANDMARKS, 600, 1, True), _dataset(Tag.POSE_WITH_LANDMARKS, 400, 2, True),
             _dataset(Tag.POSE_WITH_LANDMARKS, 100, 3, True), _dataset(Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS, 100, 4, False)]
    src = {}
    for _, f in specs:
        for i, fid in enumerate(f["frame_id"]):
            src[int(fid)] = (f, i)
    frames = [ResidentFrames(tag, {k: torch.from_numpy(v).to(DEV) for k, v in f.items()}) for tag, f in specs]
    rec = _RecordingParams()
    loader = ResidentLoader(frames, [60000.0, 40000.0, 10000.0, 10000.0], batchsize=512, steps_per_epoch=1, seed=11,
                            crop=GpuFocusRoiAugment(N, whiten=True, make_params=rec))
    (batches,) = list(loader)
    assert sum(b.meta.batchsize for b in batches) == 512 and len(batches) == 2  # split by Tag
    assert {b.meta.tag for b in batches} == {Tag.POSE_WITH_LANDMARKS, Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS}
    assert len(rec.calls) == len(batches)
    nm = A.normalization(N)
    checked_pixels = 0
    for b, prm in zip(batches, rec.calls):
        n = b.meta.batchsize
        assert b["image"].shape == (n, 1, N, N) and b["image"].dtype == torch.float32
        fid = b["frame_id"].cpu().numpy()
        rows = [src[int(x)] for x in fid]
        gather = lambda k: np.stack([f[k][i] for f, i in rows])
        scales, angles, trans = (t.cpu().numpy() for t in (prm.scales, prm.angles, prm.translations))
        # ---- INTEGER bookkeeping: bit-exact for every sample
        view = A.round_view_roi(A.compute_view_roi(gather("roi"), scales, trans, 0.3))
        assert np.array_equal(b.view_roi.cpu().numpy(), view)
        tr = A.crop_transform(view, angles.astype(np.float64), N)
        np.testing.assert_allclose(b.transform.cpu().numpy(), tr, rtol=2e-5, atol=2e-4)
        # ---- labels: crop transform, then pixel -> [-1,1] (normalize_batch)
        full = np.einsum("ij,bjk->bik", nm[:, :2].astype(np.float64), tr.astype(np.float64))
        full[:, :, 2] += nm[:, 2]
        full = full.astype(np.float32)
        np.testing.assert_allclose(b["coord"].cpu().numpy(), A.transform_coord(full, gather("coord")), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(b["pose"].cpu().numpy(), A.transform_rot(full, gather("pose")), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(b["roi"].cpu().numpy(), A.transform_roi(full, gather("roi")), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(b["pt3d_68"].cpu().numpy(), A.transform_keypoints(full, gather("pt3d_68")), rtol=1e-4, atol=3e-5)
        if "shapeparam" in b:
            assert np.array_equal(b["shapeparam"].cpu().numpy(), gather("shapeparam"))  # passes through untouched
        # ---- pixels: every 8th sample against the numpy bilinear warp (grey levels; fp32 gather arithmetic)
        crop = (b["image"].cpu().numpy()[:, 0] + 0.5) * 256.0
        imgs = gather("image")[:, 0].astype(np.float32)
        for i in range(0, n, 8):
            np.testing.assert_allclose(crop[i], A.warp_bilinear(imgs[i], tr[i], N), atol=6e-2)
            checked_pixels += 1
    assert checked_pixels >= 60
