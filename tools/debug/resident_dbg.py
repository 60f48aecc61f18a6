import sys, torch
sys.path.insert(0, "neuralnet-tracker-traincode_amd")
from trackertraincode.datasets.batch import Batch, Metadata
from trackertraincode.datatransformation.gpu import GpuFocusRoiAugment
g = torch.Generator().manual_seed(0)
n=64
f = {"image": torch.randint(0, 256, (n, 1, 96, 96), generator=g, dtype=torch.uint8),
     "roi": torch.tensor([20.0, 20.0, 76.0, 76.0]) + torch.rand(n, 4, generator=g) * 4,
     "coord": torch.cat((48 + torch.randn(n, 2, generator=g), 25 + torch.rand(n, 1, generator=g)), -1)}
b = Batch(Metadata(96, n), {k: v.cuda() for k, v in f.items()})
aug = GpuFocusRoiAugment(whiten=False)
out = aug(b, generator=torch.Generator().manual_seed(3))
c = out["coord"]; bad = c.abs().max(dim=1).values > 3
print("bad", bad.sum().item())
print(out.view_roi[bad][:5]); print(f["roi"][bad.cpu()][:5]); print(c[bad][:5]); print(out.transform[bad][:3])
print(out.view_roi[~bad][:3]); print(c[~bad][:3])
