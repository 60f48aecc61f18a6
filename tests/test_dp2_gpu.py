"""Two data-parallel replicas of the REAL training step on one GPU (BASELINE config 4's semantics minus xGMI).

Two freshly spawned processes (spawned before anything touches the GPU in them) share the box's one MI355X and exchange
gradients over the `gloo` backend: real backward hooks, real gradient arenas, GradAllReduce's buckets and in-place
all-reduces, `finish()`, 1/world folded into the fused clip+Adam (`grad_scale`).  Rank 0 compares the result with two
single-replica backward passes whose gradients it averages by hand (tests/_dp2_worker.py).  TTK_DETERMINISTIC=1 makes the
gradients of a replica reproducible, so the exchanged sums must equal the hand-made ones to rounding of the all-reduce
(none: a sum of two) and the post-step state must agree.  The 2-rank gloo test on CPU (tests/test_parallel_gloo.py) drives
the reducer with synthetic arenas; this one drives it with the real backward."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_two_replicas_real_step_equals_averaged_single_replica_gradients():
    port, world, B = _free_port(), 2, 48
    env = dict(os.environ, TTK_DETERMINISTIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_dp2_worker.py"), REPO, str(r), str(world), port, str(B)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=900)
            assert p.returncode == 0, e[-3000:]
            outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    r0 = next(o for o in outs if o["rank"] == 0)
    print(f"exchange: {r0['mode']}; {r0['collectives']} collectives for {r0['n_params']} parameter tensors, zero-copy {r0['zero_copy']} / copied {r0['copied']}; "
          f"summed gradients vs hand-made sum {r0['grad_sum_rel']:.1e}; post-step parameters {r0['param_abs']:.1e} abs ({r0['worst_key']}), "
          f"buffers {r0['buffers_rel']:.1e}; bitwise {r0['bitwise']}")
    for o in outs:
        assert o["replicas_in_sync"], outs          # both replicas hold the same parameters after the step
        assert o["copied"] == 0 and o["zero_copy"] >= 90, o  # the tensors autograd installed ARE the arena views that travelled
        assert 3 <= o["collectives"] <= 20, o        # O(buckets), not O(parameters)
    assert r0["grad_sum_rel"] < 1e-6, r0            # the all-reduced arenas hold g0 + g1
    assert r0["buffers_rel"] < 1e-6, r0             # per-replica BatchNorm statistics: rank 0's own crops only
    # Adam's first update is -lr * g / (|g| + eps): identical gradients give identical parameters; the bound for an element
    # whose gradient is rounding noise around zero would be 2 * lr
    assert r0["param_abs"] <= 1e-7 or r0["bitwise"], r0
