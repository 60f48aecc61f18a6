"""A/B on one box, fp32 path: the wide layers' weight-gradient GEMMs on a second stream beside the data-gradient chain (1) or in stream order (0, the product).
python tools/exp/ab_wgrad_stream.py <0|1> [bench.py arguments]"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode.backbones.mobilenet_v1 as MB  # noqa: E402

MB._USE_WGRAD_STREAM = bool(int(sys.argv[1]))
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
