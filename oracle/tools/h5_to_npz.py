#!/opt/conda/bin/python3.9
"""Fixture extraction (runs ONLY in the build container, with /opt/conda/bin/python3.9 + h5py 3.3).

Converts the two HDF5 data files the reference ships into .npz so that neither the oracle nor the
tests need h5py (absent from the default interpreter and from the GPU box):

  /root/reference/trackertraincode/facemodel/shapeparams_gmm.h5  -> tests/golden/shapeparams_gmm.npz
      (10-component diagonal GMM over the 50 3DMM shape parameters; read by
       trackertraincode/neuralnets/losses.py:100-104 through modelcomponents.py:246-257)
  /root/reference/aflw2kmini.h5                                  -> tests/golden/aflw2kmini.npz
      (16 AFLW2000-3D samples: JPEG byte blobs + coords/quats/rois/pt3d_68/shapeparams;
       used by test/test_landmarks.py:26-29)

These are DATA files (inputs), not reference source.

    h5_to_npz.py --dataset <in.h5> <out.npz>     converts any pose dataset of the reference's $DATADIR into a shard that
                                                 trackertraincode.pipelines.make_pose_estimation_loaders(datadir=...) reads
"""
import sys
import numpy as np
import h5py


def convert_pose_dataset(src, dst):
    """One HDF5 pose dataset of the reference (format: readme.md:214-244; reader: datasets/dshdf5pose.py:198-256) -> one .npz shard
    for trackertraincode.datasets.shards: JPEG blobs untouched (image_bytes + image_lengths) or raw frames (images), and the label
    arrays of the reader's whitelist under their HDF5 names."""
    with h5py.File(src, "r") as f:
        d = {}
        imgs = f["images"] if "images" in f else f["keys"]
        if imgs.dtype.kind == "O" or imgs.ndim == 1:  # variable-length byte blobs (storage: image_filename / jpeg)
            blobs = [np.asarray(imgs[i]).astype(np.uint8) for i in range(imgs.shape[0])]
            d["image_lengths"] = np.array([len(b) for b in blobs], dtype=np.int64)
            d["image_bytes"] = np.concatenate(blobs)
        else:
            d["images"] = imgs[...]
        for k in ["coords", "quats", "rois", "pt3d_68", "pt2d_68", "shapeparams", "hasface"]:  # the reader's label whitelist (dshdf5pose.py:168-180)
            if k in f:
                d[k] = f[k][...]
        for k in ["sequence_starts", "individual"]:  # who is in a frame (dshdf5pose.py:221-228): the "opal23" alignment of the evaluation needs it
            if k in f:
                d[k] = f[k][...]
        np.savez(dst, **d)
        print(dst, {k: (v.shape, v.dtype) for k, v in d.items()})


if len(sys.argv) > 1 and sys.argv[1] == "--dataset":
    convert_pose_dataset(sys.argv[2], sys.argv[3])
    sys.exit(0)

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = sys.argv[2] if len(sys.argv) > 2 else "/root/repo/tests/golden"

with h5py.File(f"{ref}/trackertraincode/facemodel/shapeparams_gmm.h5", "r") as f:
    assert f.attrs["covariance_type"] == "diag"
    np.savez(
        f"{out}/shapeparams_gmm.npz",
        weights=f["weights"][...],
        means=f["means"][...],
        cov=f["cov"][...],
    )
    print("gmm", f["weights"].shape, f["means"].shape, f["cov"].shape, f["weights"].dtype)

with h5py.File(f"{ref}/aflw2kmini.h5", "r") as f:
    def show(name, obj):
        if isinstance(obj, h5py.Dataset):
            print(name, obj.shape, obj.dtype, dict(obj.attrs))
    f.visititems(show)
    d = {}
    imgs = f["images"]
    blobs = [np.asarray(imgs[i]).astype(np.uint8) for i in range(imgs.shape[0])]
    d["image_lengths"] = np.array([len(b) for b in blobs], dtype=np.int64)
    d["image_bytes"] = np.concatenate(blobs)
    for k in ["coords", "quats", "rois", "pt3d_68", "shapeparams"]:
        d[k] = f[k][...]
    np.savez(f"{out}/aflw2kmini.npz", **d)
    print({k: (v.shape, v.dtype) for k, v in d.items()})
