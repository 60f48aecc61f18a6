"""Throughput of the training loader alone (draw -> gather -> crop/warp -> intensity augmentation) with the frames in HBM and in pinned host
memory (datasets/resident.py), on synthetic frames: python tools/loader_bench.py [--frames 20000] [--size 256] [--batch 512] [--steps 60]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neuralnet-tracker-traincode_amd"))
from trackertraincode.datasets.resident import ResidentFrames, ResidentLoader  # noqa: E402
from trackertraincode.datatransformation.gpu import GpuFocusRoiAugment  # noqa: E402
from trackertraincode.pipelines import Tag, make_image_augmentations  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=20000)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=60)
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
N, S = a.frames, a.size
fields = {
    "image": torch.randint(0, 255, (N, 1, S, S), dtype=torch.uint8, generator=g),
    "roi": torch.tensor([[0.2 * S, 0.2 * S, 0.8 * S, 0.8 * S]]).repeat(N, 1) + torch.randn(N, 4, generator=g) * 4,
    "coord": torch.tensor([[0.5 * S, 0.5 * S, 0.25 * S]]).repeat(N, 1),
    "pose": torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1),
    "pt3d_68": torch.rand(N, 68, 3, generator=g) * S,
    "shapeparam": torch.randn(N, 50, generator=g),
    "coord_convention_id": torch.zeros(N, dtype=torch.int32),
}
host = ResidentFrames(Tag.POSE_WITH_LANDMARKS, fields)
for placement in ("device", "host"):
    frames = host.to("cuda") if placement == "device" else host.to_host()
    augs = make_image_augmentations(torch.Generator().manual_seed(1))
    crop = GpuFocusRoiAugment(new_size=129, rotation_aug_angle=30.0, extension_factor=1.1, whiten=False, flip_rot_p=0.01)
    loader = ResidentLoader([frames], [1.0], a.batch, a.steps, seed=3, crop=crop, image_augmentations=augs)
    for _ in loader:  # warm-up epoch (allocator, pinned staging buffers)
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for step in loader:
        n += sum(int(b["image"].shape[0]) for b in step)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"frames on {placement:6s}: {n / dt:10.0f} crops/s  ({dt / a.steps * 1e3:.2f} ms per batch of {a.batch}, source frames {S}x{S}, {frames.nbytes() / 2**30:.2f} GiB)", flush=True)
