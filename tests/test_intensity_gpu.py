"""GPU parity of the fused intensity-augmentation kernel (C-ABI ttk_intensity_augment) against oracle/intensity.py on
explicit per-sample parameters, and statistics of the host-side sampling.  kornia is absent: PARITY UNPINNED (the oracle
restates kornia's published formulas; see its header)."""
import numpy as np
import pytest
import torch

from oracle import intensity as O


def _images(B, H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    img = np.empty((B, H, W), np.float32)
    for n in range(B):
        base = 0.5 + 0.35 * np.sin(xx * rng.uniform(0.02, 0.2) + yy * rng.uniform(0.02, 0.2) + rng.uniform(0, 6))
        img[n] = np.clip(base * rng.uniform(0.3, 1.0) + rng.normal(0, 0.05, (H, W)), 0, 1)
    img[0, :4, :4] = 1.0  # exercises the histogram's last bin
    return np.floor(img * 255.0).astype(np.float32) / np.float32(256.0)  # what normalize_batch hands over: u8 / 256


def _params(B, seed):
    rng = np.random.default_rng(seed)
    p = np.zeros((B, O.NPARAMS), np.float32)
    p[:, O.EQUALIZE] = rng.random(B) < 0.5
    p[:, O.POSTERIZE_BITS] = np.where(rng.random(B) < 0.4, rng.integers(1, 8, B), 0)
    p[:, O.GAMMA] = np.where(rng.random(B) < 0.5, rng.uniform(0.5, 2.0, B), 0)
    p[:, O.CONTRAST] = np.where(rng.random(B) < 0.5, rng.uniform(0.7, 1.5, B), 0)
    p[:, O.BRIGHTNESS] = np.where(rng.random(B) < 0.5, rng.uniform(0.7, 1.5, B), 0)
    p[:, O.BLUR] = rng.random(B) < 0.4
    p[:, O.NOISE_STD] = np.where(rng.random(B) < 0.5, rng.choice([4, 16, 32, 64], B) / 255.0, 0)
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(129, 129), (31, 47)])
def test_kernel_matches_oracle_on_explicit_parameters(H, W):
    import trackertraincode._hip as hip
    L, p = hip.lib(), hip.ptr
    B = 48
    x, prm = _images(B, H, W, 3), _params(B, 4)
    prm[:8] = 0  # single operations in isolation
    for i, slot in enumerate([O.EQUALIZE, O.POSTERIZE_BITS, O.GAMMA, O.CONTRAST, O.BRIGHTNESS, O.BLUR, O.NOISE_STD]):
        prm[i, slot] = {O.POSTERIZE_BITS: 4, O.GAMMA: 0.6, O.CONTRAST: 1.4, O.BRIGHTNESS: 0.8, O.NOISE_STD: 16 / 255.0}.get(slot, 1.0)
    noise = np.random.default_rng(5).standard_normal((B, H, W)).astype(np.float32)
    ref = O.augment(x, prm, noise, out_shift=-0.5)
    dx, dp, dn = torch.from_numpy(x).cuda(), torch.from_numpy(prm).cuda(), torch.from_numpy(noise).cuda()
    y = torch.empty_like(dx)
    L.call("ttk_intensity_augment", p(dx), p(y), p(dp), p(dn), B, H, W, -0.5)
    got = y.cpu().numpy()
    err = np.abs(got - ref)
    # tolerance 2e-6 (fp32 pow / blur summation order).  Posterize after gamma/equalize quantises: a 1-ulp difference of
    # pow can move a pixel across a level - those few pixels are allowed a whole level.
    bad = err > 2e-6
    assert bad.mean() < 1e-4, f"{bad.sum()} of {bad.size} pixels differ (max {err.max():.3e})"
    for n in range(8):
        assert np.abs(got[n] - ref[n]).max() <= 2e-6, f"single operation {n}"
    assert got.min() >= -0.5 and got.max() <= 0.5
    # in place, no noise tensor: noise parameter ignored
    L.call("ttk_intensity_augment", p(dx), p(dx), p(dp), None, B, H, W, 0.0)
    ref0 = O.augment(x, prm, None, 0.0)
    assert (np.abs(dx.cpu().numpy() - ref0) > 2e-6).mean() < 1e-4
    with pytest.raises(RuntimeError, match="LDS"):
        L.call("ttk_intensity_augment", p(dx), p(dx), p(dp), None, 1, 200, 200, 0.0)


@pytest.mark.gpu
def test_container_on_batch_and_noise_statistics():
    import trackertraincode.datatransformation as dtr
    from trackertraincode.datasets.batch import Batch, Metadata
    from trackertraincode.datatransformation.tensors.affinetrafo import FieldCategory

    g = torch.Generator().manual_seed(0)
    noise_only = dtr.batch.KorniaImageDistortions(dtr.batch.RandomGaussianNoise(std=16.0 / 255.0, p=1.0), dtr.batch.OnlyClip(p=1.0), generator=g)
    img = torch.full((64, 1, 129, 129), 0.5, device="cuda")
    meta = Metadata(129, 64, categories={"image": FieldCategory.image})
    out = noise_only(Batch(meta, {"image": img, "roi": torch.zeros(64, 4, device="cuda")}))
    assert out["image"].shape == img.shape and out["roi"].shape == (64, 4)
    d = (out["image"] - 0.5).flatten()
    assert abs(d.mean().item()) < 2e-4 and abs(d.std().item() - 16.0 / 255.0) < 3e-4
    with pytest.raises(RuntimeError, match="CUDA"):
        noise_only.apply(torch.zeros(2, 1, 8, 8), torch.zeros(2, 8))


def test_oracle_properties():
    """Known-answer checks of the restated formulas (CPU)."""
    x = _images(4, 33, 29, 1)
    flat = np.full((1, 8, 8), 0.25, np.float32)
    assert np.array_equal(O.equalize(flat[0]), flat[0])  # one populated bin: step 0, unchanged
    ramp = (np.arange(256, dtype=np.float32).reshape(16, 16) / np.float32(255.0))
    eq = O.equalize(np.tile(ramp, (4, 4)))  # uniform histogram maps (almost) onto itself
    assert np.abs(eq - np.tile(ramp, (4, 4))).max() <= 1.0 / 255.0 + 1e-6
    e = O.equalize(x[1])
    assert e.min() >= 0 and e.max() <= 1 and np.unique(np.round(e * 255)).size <= 256
    order = np.argsort(x[1].ravel(), kind="stable")
    assert np.all(np.diff(e.ravel()[order]) >= 0)  # a monotone grey-level map
    assert e.std() > x[1].std() and abs(np.median(e) - 0.5) < 0.15  # spreads the histogram around mid-grey
    pz = O.posterize(x[0], 4)
    assert set(np.unique(np.round(pz * 255).astype(int)) % 16) == {0}
    assert np.array_equal(O.posterize(x[0], 8), x[0])
    assert np.allclose(O.adjust_gamma(x[0], 1.0), x[0], atol=1e-7)
    assert np.allclose(O.adjust_brightness(x[0], 1.0), x[0]) and np.allclose(O.adjust_contrast(x[0], 1.0), x[0])
    assert O.adjust_brightness(x[0], 1.5).max() <= 1.0
    b = O.gaussian_blur5(np.full((9, 11), 0.3, np.float32))
    assert np.allclose(b, 0.3, atol=1e-6)  # normalised kernel, reflect border keeps constants
    imp = np.zeros((9, 9), np.float32)
    imp[4, 4] = 1
    k = O.gaussian_blur5(imp)
    assert abs(k.sum() - 1) < 1e-6 and np.allclose(k, k.T) and k[4, 4] == k.max() and k[4, 1] == 0
    edge = np.zeros((9, 9), np.float32)
    edge[0, 0] = 1
    ke = O.gaussian_blur5(edge)  # reflect (no edge repeat): column 0 receives weight from column 0 only
    g1 = np.exp(-np.arange(-2, 3) ** 2 / 4.5)
    g1 /= g1.sum()
    assert abs(ke[0, 0] - g1[2] ** 2) < 1e-6 and abs(ke[1, 1] - g1[1] ** 2 - 0 * g1[3]) < 1e-6 + g1[3] ** 2 * 2


def test_parameter_sampling_statistics():
    """Host-side sampling: firing rates, ranges, random_apply subset size, noise ladder root-sum-square."""
    import trackertraincode.datatransformation as dtr
    B = dtr.batch
    g = torch.Generator().manual_seed(1)
    chain = B.KorniaImageDistortions(B.RandomEqualize(p=0.2), B.RandomPosterize((4.0, 6.0), p=0.01), B.RandomGamma((0.5, 2.0), p=0.2),
                                     B.RandomContrast((0.7, 1.5), p=0.2), B.RandomBrightness((0.7, 1.5), p=0.2),
                                     B.RandomGaussianBlur(p=0.1, kernel_size=(5, 5), sigma=(1.5, 1.5)), random_apply=4, generator=g)
    n, rounds = 4096, 60
    fired = np.zeros(6)
    for _ in range(rounds):
        prm = chain.sample_params(n).numpy()
        active = (prm[:, :6] > 0).any(axis=0)
        assert active.sum() <= 4  # at most 4 of the 6 operations per call
        fired += (prm[:, :6] > 0).mean(axis=0)
        gm = prm[:, 2][prm[:, 2] > 0]
        assert gm.size == 0 or (gm.min() >= 0.5 and gm.max() <= 2.0)
        bits = prm[:, 1][prm[:, 1] > 0]
        assert set(bits.tolist()) <= {4.0, 5.0, 6.0}
    rate = fired / rounds  # expected p * 4/6
    expect = np.array([0.2, 0.01, 0.2, 0.2, 0.2, 0.1]) * 4 / 6
    assert np.all(np.abs(rate - expect) < 0.35 * expect + 0.004), (rate, expect)
    ladder = B.KorniaImageDistortions(B.RandomGaussianNoise(std=4 / 255, p=0.25), B.RandomGaussianNoise(std=16 / 255, p=0.25 ** 2),
                                      B.RandomGaussianNoise(std=32 / 255, p=0.25 ** 3), B.RandomGaussianNoise(std=64 / 255, p=0.25 ** 4),
                                      B.OnlyClip(p=1.0), generator=g)
    s = ladder.sample_params(200000).numpy()[:, 6]
    assert abs((s > 0).mean() - (1 - 0.75 * (1 - 1 / 16) * (1 - 1 / 64) * (1 - 1 / 256))) < 0.004
    levels = np.unique(np.round(s[s > 0] * 255, 3))
    assert np.abs(levels - 4.0).min() < 1e-3 and np.abs(levels - 16.0).min() < 1e-3 and np.abs(levels - np.hypot(4, 16)).min() < 2e-3
    with pytest.raises(NotImplementedError):
        B.KorniaImageDistortions(B.RandomGamma((0.5, 2.0)), B.RandomEqualize())  # not the kernel's order


@pytest.mark.gpu
def test_pipeline_loader_applies_augmentations():
    from trackertraincode import pipelines

    train, _, _ = pipelines.make_pose_estimation_loaders(129, 64, "synthetic", enable_image_aug=True, device="cuda")
    plain, _, _ = pipelines.make_pose_estimation_loaders(129, 64, "synthetic", enable_image_aug=False, device="cuda")
    a, b = next(iter(train)), next(iter(plain))
    ia, ib = torch.cat([x["image"] for x in a]), torch.cat([x["image"] for x in b])
    assert ia.shape == ib.shape and ia.min() >= -0.5 and ia.max() <= 0.5
    changed = ((ia - ib).abs().flatten(1).max(dim=1).values > 1e-6).float().mean().item()
    assert 0.2 < changed < 0.95  # most samples get at least one operation, some none


@pytest.mark.gpu
def test_resident_loader_feeds_training_step():
    """HBM-resident uint8 frames -> weighted draw -> HIP crop/warp + intensity augmentation -> list[Batch] -> one step."""
    import importlib.util, os
    import trackertraincode.train as train
    from trackertraincode import pipelines
    from trackertraincode.datasets.resident import ResidentFrames, ResidentLoader
    from trackertraincode.neuralnets.models import NetworkWithPointHead
    from trackertraincode.pipelines import Tag

    g = torch.Generator().manual_seed(0)
    dev = "cuda"

    def frames(n, tag, with_shape):
        f = {"image": torch.randint(0, 256, (n, 1, 96, 96), generator=g, dtype=torch.uint8),
             "roi": torch.tensor([20.0, 20.0, 76.0, 76.0]) + torch.rand(n, 4, generator=g) * 4,
             "coord": torch.cat((48 + torch.randn(n, 2, generator=g), 25 + torch.rand(n, 1, generator=g)), -1),
             "pose": torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1),
             "pt3d_68": 48 + 10 * torch.randn(n, 68, 3, generator=g),
             "coord_convention_id": torch.zeros(n, dtype=torch.int32)}
        if with_shape:
            f["shapeparam"] = torch.randn(n, 50, generator=g) * 0.5
        return ResidentFrames(tag, {k: v.to(dev) for k, v in f.items()})

    loader = ResidentLoader([frames(300, Tag.POSE_WITH_LANDMARKS, True), frames(40, Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS, False)],
                            [11.0, 1.0], batchsize=64, steps_per_epoch=2, seed=1,
                            image_augmentations=pipelines.make_image_augmentations(torch.Generator().manual_seed(3)))
    steps = list(loader)
    assert len(steps) == 2
    for batches in steps:
        assert sum(b.meta.batchsize for b in batches) == 64
        for b in batches:
            n = b.meta.batchsize
            assert b["image"].shape == (n, 1, 129, 129) and b["image"].dtype == torch.float32
            assert b["image"].min() >= -0.5 and b["image"].max() <= 0.5
            assert b["pt3d_68"].shape == (n, 68, 3) and b["coord"].abs().max() < 3 and ("shapeparam" in b) == (b.meta.tag == Tag.POSE_WITH_LANDMARKS)
            assert torch.allclose(b["pose"].norm(dim=-1), torch.ones(n, device=dev), atol=1e-5)
    spec = importlib.util.spec_from_file_location("amd_train_script", os.path.join(os.path.dirname(__file__), "..", "neuralnet-tracker-traincode_amd", "scripts", "train_poseestimator.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    args = S.make_parser().parse_args([])
    args.with_pointhead, args.with_nll_loss, args.rampup_nll_losses = True, False, False
    torch.manual_seed(0)
    net = NetworkWithPointHead(enable_point_head=True, enable_uncertainty=False, config="mobilenetv1", backbone_args={"use_blurpool": False}).to(dev).train()
    crit, _ = S.setup_losses(args, net)
    out = train.training_step(net, steps[0], 0, crit)
    out["loss"].backward()
    assert torch.isfinite(out["loss"]) and all(torch.isfinite(q.grad).all() for q in net.parameters() if q.grad is not None)


@pytest.mark.gpu
def test_resident_loader_same_tag_datasets_of_different_frame_sizes():
    """The reference's baseline mix (--ds repro_300_wlp+lapa_megaface_lp:20000+wflw_lp) has three datasets of ONE Tag; decoded shards are
    padded to their own largest frame, so the parts of a step cannot be stacked before the crop (round-3 advisor finding: torch.cat
    raised).  Each part is cropped on its own and the 129 x 129 crops are collated per Tag; every sample still equals the crop its own
    dataset would produce alone (deterministic crop: no roi randomisation, no intensity augmentation)."""
    from trackertraincode.datasets.resident import ResidentFrames, ResidentLoader
    from trackertraincode.datatransformation.batch.geometric import NoRoiRandomization
    from trackertraincode.datatransformation.gpu import GpuFocusRoiAugment
    from trackertraincode.pipelines import Tag

    g = torch.Generator().manual_seed(4)
    dev = "cuda"

    def frames(n, h, w):
        f = {"image": torch.randint(0, 256, (n, 1, h, w), generator=g, dtype=torch.uint8),
             "roi": torch.tensor([10.0, 12.0, min(w, h) - 8.0, min(w, h) - 6.0]) + torch.rand(n, 4, generator=g) * 3,
             "coord": torch.cat((w / 2 + torch.randn(n, 2, generator=g), 20 + torch.rand(n, 1, generator=g)), -1),
             "pose": torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1),
             "pt3d_68": w / 2 + 8 * torch.randn(n, 68, 3, generator=g),
             "coord_convention_id": torch.zeros(n, dtype=torch.int32),
             "index": torch.arange(n, dtype=torch.int32)}
        return ResidentFrames(Tag.POSE_WITH_LANDMARKS, {k: v.to(dev) for k, v in f.items()})

    sets = [frames(40, 64, 64), frames(30, 96, 80), frames(20, 72, 112)]
    crop = GpuFocusRoiAugment(make_params=NoRoiRandomization(1.1), whiten=True)
    loader = ResidentLoader(sets, [2.0, 1.0, 1.0], batchsize=48, steps_per_epoch=3, seed=2, crop=crop)
    seen = 0
    for batches, plan in zip(loader, _replay_draws(sets, [2.0, 1.0, 1.0], 48, 3, seed=2)):
        assert len(batches) == 1 and batches[0].meta.batchsize == 48  # one Tag -> one Batch
        b = batches[0]
        assert b["image"].shape == (48, 1, 129, 129)
        # the same frames through the crop of their own dataset alone, in the loader's order (datasets in first-seen order of the draw)
        lo = 0
        for d, idx in plan:
            sel = torch.from_numpy(idx).to(dev)
            data = {k: v.index_select(0, sel) for k, v in sets[d].fields.items()}
            from trackertraincode.datasets.batch import Batch, Metadata
            from trackertraincode.datasets.resident import _CATEGORIES
            ref = crop(Batch(Metadata(tuple(data["image"].shape[-2:][::-1]), len(idx), sets[d].tag, None, {k: c for k, c in _CATEGORIES.items() if k in data}), data))
            n = len(idx)
            for k in ("image", "coord", "pose", "pt3d_68", "roi", "index"):
                assert torch.equal(b[k][lo:lo + n], ref[k]), (k, d)
            lo += n
            seen += n
        assert lo == 48
    assert seen == 3 * 48


def _replay_draws(sets, weights, batchsize, steps, seed):
    """The (dataset, frame indices) plan of a ResidentLoader with the same seed (its draw() is deterministic)."""
    from trackertraincode.datasets.resident import ResidentLoader
    twin = ResidentLoader(sets, weights, batchsize=batchsize, steps_per_epoch=steps, seed=seed)
    return [twin.draw() for _ in range(steps)]
