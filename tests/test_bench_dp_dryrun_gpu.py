"""bench.py's multi-rank code path (rendezvous, parameter broadcast, in-place gradient all-reduce during backward, barrier + max-over-ranks
timing, rank-0 JSON line) on the one GPU this pool offers: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ...`), sharing device 0 over gloo.  RCCL needs one device per rank, so the
collective library differs from the measured configuration - the Python around it does not.  The rate is not a result."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [["--comm-only"], ["--backbone", "resnet18"], ["--precision", "bf16-compute"], ["--blurpool"]],
                         ids=["default+comm-only", "resnet18", "bf16-compute", "blurpool"])
def test_two_ranks_sharing_the_gpu(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29571",
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "64", "--dist-backend", "gloo", "--share-gpu",
           "--no-cpu-baseline", "--no-copy-probe", *extra]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak" and d["dtype"] == ("bf16" if "--precision" in extra else "f32")
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2" and "dry run" in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - 128 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["roofline"] is not None and d["roofline"]["traffic"] is None  # B = 64 per rank: not the default workload, no counters attached
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None
    # the diagnostics of the exchange (round 5): both ranks were seen, the reducer issued a handful of buckets that carry every gradient
    c = d["comm"]
    assert c["world_seen"] == 2 and c["backend"] == "gloo" and 1 <= c["buckets"] <= 40
    n_param_bytes = {"--backbone": 11_000_000 * 4}.get(extra[0] if extra else "", 3_200_000 * 4)
    assert c["bytes_per_step"] >= n_param_bytes and c["exposed_ms"] >= 0.0 and c["allreduce_ms_sum"] > 0.0
    if "--comm-only" in extra:
        co = c["comm_only"]
        assert co["ms_per_step"] > 0 and co["bytes"] == c["bytes_per_step"] and co["buckets"] == c["buckets"]
