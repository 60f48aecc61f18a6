"""Teacher-forced K-step walk (tests/_teacher_forced_worker.py): for t = 0 .. 9 the HIP model and the fused clip + Adam are loaded with the
fp32 oracle's COMPLETE state at step t (parameters, BatchNorm buffers, Adam moments, step counts), make one step of the real loop, and
must reproduce the oracle's step t -> t + 1 with the single-step tolerances - at the full learning rate 1e-3, B = 256 (BASELINE config 2),
default and full-uncertainty configurations.  This pins what a free walk cannot (tests/test_trajectory_gpu.py: two fp32 walks part
chaotically after Adam's first, sign-like updates): optimiser-state and running-statistics evolution at EVERY t, on states the training
actually visits (moments that are no longer zero, statistics that have moved).
Reference loop: trackertraincode/train.py:372-439, scripts/train_poseestimator.py:147-167,442-454.

Tolerances (north_star: per-step losses within 1e-3):
  loss_sum 1e-4; every per-sample loss 1e-3; BatchNorm running statistics 2e-4 (relative, as tests/test_fullsize_gpu.py);
  global gradient norm 1e-3 relative;
  Adam moments: from identical moments the new ones differ by (1 - beta) x the difference of the clipped gradients, and a gradient tensor of
  this network carries 2-7e-3 (L2, relative) of fp32 rounding noise in EITHER fp32 implementation (tests/test_fullsize_gpu.py: both are
  that far from float64).  So: at the last step every moment tensor is held to the float64 oracle's step from the same state as closely
  as the fp32 oracle is (3 x its distance + 1e-5, per tensor, L2) - the yardstick criterion of the single-step tests - and at every step
  exp_avg within 2e-2 / exp_avg_sq within 3e-2 (per tensor, L2, against the fp32 oracle; measured 3-7e-3 / 2-11e-3 for the worst tensor)
  and exp_avg within 1e-4 absolute (measured 3e-5);
  parameters after the update: never further than 2 lr from the oracle's (an element whose gradient is rounding noise around zero moves
  +-lr in either implementation: Adam's update is lr * m / (sqrt(v) + eps)) - printed, and asserted as that bound."""
import os

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 10


@pytest.mark.parametrize("cfg", ["default", "full"])
def test_ten_teacher_forced_steps_at_the_full_learning_rate(cfg, walk_workers):
    r = walk_workers.result(("teacher", cfg, 256, K, 150))  # (tests/conftest.py: the walks of this session run side by side)
    f = lambda xs: "[" + " ".join("%.1e" % x for x in xs) + "]"
    print(f"cfg={cfg} lr={r['lr']:.3g}: loss oracle {['%.5f' % x for x in r['loss_oracle']]}\n  |dloss| {f(r['dloss'])}\n  per-sample {f(r['dsample'])}\n"
          f"  running stats {f(r['running_rel'])}\n  grad norm rel {f(r['dgnorm_rel'])}\n  exp_avg rel {f(r['m_rel'])} abs {f(r['m_abs'])} ({r['worst_m'][-1]})\n"
          f"  exp_avg_sq rel {f(r['v_rel'])}\n  per tensor L2: exp_avg worst {f(r['m_l2_worst'])} median {f(r['m_l2_median'])}; exp_avg_sq worst {f(r['v_l2_worst'])} "
          f"median {f(r['v_l2_median'])}\n  parameters after the update / lr {f(r['param_over_lr'])}")
    print(f"  moments against the float64 step at t = {K - 1}: largest (hip distance) / (3 x cpu32 distance + 1e-5) = {r['yard_ratio']:.2f} ({r['yard_worst']})")
    assert r["yard_ratio"] <= 1.0, (r["yard_ratio"], r["yard_worst"])
    assert abs(r["lr"] - 1.0e-3) < 1e-12
    assert r["loss_oracle"][-1] < r["loss_oracle"][0]  # the teacher walks downhill: later states differ from the initial one
    for t in range(K):
        assert r["dloss"][t] < 1e-4, (t, r["dloss"])
        assert r["dsample"][t] < 1e-3, (t, r["dsample"])
        assert r["running_rel"][t] < 2e-4, (t, r["running_rel"], r["worst_running"][t])
        assert r["dgnorm_rel"][t] < 1e-3, (t, r["dgnorm_rel"])
        assert r["m_l2_worst"][t] < 2e-2 and r["m_abs"][t] < 1e-4, (t, r["m_l2_worst"], r["m_abs"], r["worst_m"][t])
        assert r["v_l2_worst"][t] < 3e-2, (t, r["v_l2_worst"])
        assert r["param_over_lr"][t] <= 2.02, (t, r["param_over_lr"])


test_ten_teacher_forced_steps_at_the_full_learning_rate.walk_job = lambda cfg: ("teacher", cfg, 256, K, 150)
