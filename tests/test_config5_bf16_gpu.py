"""BASELINE config 5 on the one GPU this pool offers: the multi-task mix (three dataset kinds with different label sets) drawn by the
weighted sampler, random focus-ROI crop + affine warp + label bookkeeping + mirror / quarter turns + intensity augmentation on the GPU,
NLL losses with their ramp, clip + Adam - in the bf16-compute mode (BASELINE config 5's bf16 leg; the storage-only bf16 variants are retired), at 512 crops per step, through train.fit().  (The 8-GPU part of
the configuration - RCCL all-reduce - is covered by tests/test_parallel_*.py and test_dp2_gpu.py.)  The bf16 run is held to the fp32 run
of the same data: same draws, same augmentation parameters; per-step losses within the tolerance measured for single steps
(tests/test_bf16_compute_gpu.py), loosened for the parameter drift of the preceding steps."""
import numpy as np
import pytest
import torch

from util import script_args, train_script

pytestmark = pytest.mark.gpu


def _frames(tag, n, size, seed, with_pts, with_shape):
    from trackertraincode.datasets.resident import ResidentFrames

    g = torch.Generator().manual_seed(seed)
    S = size
    f = {
        "image": torch.randint(0, 255, (n, 1, S, S), dtype=torch.uint8, generator=g),
        "roi": torch.tensor([[0.25 * S, 0.25 * S, 0.75 * S, 0.75 * S]]).repeat(n, 1) + torch.randn(n, 4, generator=g) * 3,
        "coord": torch.cat([torch.full((n, 2), 0.5 * S) + torch.randn(n, 2, generator=g) * 4, torch.full((n, 1), 0.22 * S)], -1),
        "pose": torch.nn.functional.normalize(torch.cat([torch.randn(n, 3, generator=g) * 0.3, torch.ones(n, 1)], -1), dim=-1),
        "coord_convention_id": torch.full((n,), seed % 3, dtype=torch.int32),
    }
    if with_pts:
        f["pt3d_68"] = torch.cat([torch.rand(n, 68, 2, generator=g) * 0.4 * S + 0.3 * S, torch.randn(n, 68, 1, generator=g) * 10], -1)
    if with_shape:
        f["shapeparam"] = torch.randn(n, 50, generator=g) * 0.5
    return ResidentFrames(tag, {k: v.cuda() for k, v in f.items()})


def _run(mode, steps=4, B=512):
    import trackertraincode.backbones.mobilenet_v1 as MB
    import trackertraincode.train as train
    from trackertraincode.datasets.resident import ResidentLoader
    from trackertraincode.datatransformation.gpu import GpuFocusRoiAugment
    from trackertraincode.pipelines import Tag, make_image_augmentations

    S = train_script()
    sets = [_frames(Tag.POSE_WITH_LANDMARKS, 700, 160, 1, True, True), _frames(Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS, 300, 128, 2, True, False),
            _frames(Tag.ONLY_POSE, 300, 192, 3, False, False)]
    augs = make_image_augmentations(torch.Generator().manual_seed(100))
    crop = GpuFocusRoiAugment(new_size=129, rotation_aug_angle=30.0, extension_factor=1.1, whiten=False, flip_rot_p=0.01)
    loader = ResidentLoader(sets, [0.6, 0.2, 0.2], B, steps, seed=9, crop=crop, image_augmentations=augs)
    flags = dict(with_pointhead=True, with_nll_loss=True, rampup_nll_losses=True)
    args = script_args(flags, epochs=2)
    torch.manual_seed(0)
    net = S.create_net(args).cuda()
    g = torch.Generator().manual_seed(7)
    net.landmarks.deformablekeypoints.set_basis(torch.randn(68, 3, generator=g) * 0.5, torch.randn(50, 68, 3, generator=g) * 0.05)
    crit, _ = S.setup_losses(args, net)
    opt, sch = S.create_optimizer(net, args)
    losses, tags = [], []

    def on_step(epoch, out):
        losses.append(float(out["loss"].detach()))

    MB.set_activation_dtype(mode)
    try:
        torch.manual_seed(3)  # the noise augmentation draws from the global device generator
        train.fit(net, loader, crit, opt, sch, epochs=1, on_step=on_step)
        torch.cuda.synchronize()
    finally:
        MB.set_activation_dtype("fp32")
    first = next(iter(ResidentLoader(sets, [0.6, 0.2, 0.2], B, 1, seed=9, crop=crop, image_augmentations=None)))
    return losses, [b.meta.tag for b in first], [int(b["image"].shape[0]) for b in first], net


@pytest.fixture(scope="module")
def fp32_run():
    return _run("fp32")


# first-step tolerance: the same parameters, only the mode's rounding differs.  bf16-compute (bf16 tensors AND one bf16 MFMA
# product per pointwise convolution): 2e-3 (measured 1e-4 at this batch)
@pytest.mark.parametrize("mode", ["bf16-compute"])
def test_multitask_mix_in_bf16_tracks_fp32(mode, fp32_run):
    from trackertraincode.pipelines import Tag

    l32, tags, sizes, net32 = fp32_run
    l16, _, _, net16 = _run(mode)
    assert set(tags) == {Tag.POSE_WITH_LANDMARKS, Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS, Tag.ONLY_POSE} and sum(sizes) == 512
    assert len(l32) == len(l16) == 4 and all(np.isfinite(l32)) and all(np.isfinite(l16))
    print("fp32 losses", l32, mode, "losses", l16)
    assert abs(l16[0] - l32[0]) <= 2e-3 * abs(l32[0])
    for a, b in zip(l32[1:], l16[1:]):
        assert abs(b - a) <= 1e-1 * abs(a)  # later steps (measured 0.5 %, 1.5 %, 4.2 % for bf16 storage): Adam's first updates amplify the gradients' bf16 noise
    assert all(torch.isfinite(p).all() for p in net16.parameters())
    moved = [float((p16.detach() - p32.detach()).abs().max()) for p16, p32 in zip(net16.parameters(), net32.parameters())]
    assert max(moved) > 0.0  # the two runs are not the same run
