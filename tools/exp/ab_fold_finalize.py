"""A/B on one box: the depthwise weight-gradient rows folded by the BatchNorm-backward finalisation's launch (product) or by their own launch.
   python tools/exp/ab_fold_finalize.py <0|1|2|3> [bench.py arguments]"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode.backbones._mobilenet_bc as BC  # noqa: E402

BC._FOLD_WITH_FINALIZE = int(sys.argv[1])  # 0 none | 1 depthwise rows | 2 + fused early layers | 3 + wide layers
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
