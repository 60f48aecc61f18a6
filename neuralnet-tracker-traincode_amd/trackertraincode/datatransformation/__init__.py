"""Label/image transforms for crops (reference: trackertraincode/datatransformation).  In scope here:
the affine bookkeeping (tensors/affinetrafo.py), normalisation, the ROI randomisation of
batch/geometric.py, and `gpu.GpuFocusRoiAugment`, the batched MI355X replacement of the per-sample
OpenCV warp.  The OpenCV / kornia / HDF5 parts of the reference's pipeline are not part of this package."""
from . import batch, tensors  # noqa: F401
from .gpu import GpuFocusRoiAugment  # noqa: F401
