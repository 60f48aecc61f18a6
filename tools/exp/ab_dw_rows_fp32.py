"""A/B on one box, fp32 path: the fused depthwise weight gradient by float atomics (0, product until round 5), by workgroup rows + their own fold launch (1),
or by rows folded inside the BatchNorm-backward finalisation's launch (2).   python tools/exp/ab_dw_rows_fp32.py <0|1|2> [bench.py arguments]"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode.backbones.mobilenet_v1 as MB  # noqa: E402

MB._DW_WGRAD_ROWS = int(sys.argv[1])
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
