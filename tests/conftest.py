import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "neuralnet-tracker-traincode_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fixed_cpu_threads():
    """The CPU oracle's fp32 rounding depends on how torch splits reductions over threads; with tiny batches one ReLU decision on the other
    side moves whole gradients by 1e-3 (tests/test_backbone_gpu.py).  A fixed thread count makes the oracle's values the same on every
    box with at least 8 cores; the large-batch tests raise it themselves for speed."""
    import torch

    torch.set_num_threads(min(8, os.cpu_count() or 1))
    yield


class _WalkWorkers:
    """The six CPU-oracle walks of tests/test_teacher_forced_gpu.py and tests/test_trajectory_gpu.py (each a worker process: 5 - 10 optimiser
    steps of the fp32 AND the float64 oracle at B = 256, 50 - 70 s of host time, a few seconds of GPU time) run side by side instead of one
    after the other: the workers of EVERY such test selected in the session start right after collection (<= 6 x 32 host threads; the boxes
    have 256) and finish under the session's other tests; a test waits for its own worker only.  Nothing about a worker changes - same script, arguments, environment, assertions."""

    def __init__(self, session):
        self._session, self._procs, self._tmp = session, {}, None

    @staticmethod
    def argv(kind, *args):
        script = {"trajectory": "_trajectory_worker.py", "teacher": "_teacher_forced_worker.py"}[kind]
        return [sys.executable, os.path.join(REPO, "tests", script), REPO, *[str(a) for a in args]]

    def _selected(self):
        jobs = []
        for item in self._session.items:
            fn = getattr(item, "function", None)
            spec = getattr(fn, "walk_job", None)
            if spec is not None and hasattr(item, "callspec"):
                jobs.append(spec(**{k: v for k, v in item.callspec.params.items()}))
        return jobs

    def _start(self, job):
        import subprocess
        import tempfile

        if job in self._procs:
            return
        if self._tmp is None:
            self._tmp = tempfile.mkdtemp(prefix="ttk_walks_")
        base = os.path.join(self._tmp, str(len(self._procs)))
        out, err = open(base + ".out", "w+"), open(base + ".err", "w+")
        env = dict(os.environ, TTK_DETERMINISTIC="1")
        self._procs[job] = (subprocess.Popen(self.argv(*job), env=env, stdout=out, stderr=err, text=True), out, err)

    def result(self, job, timeout=3000):
        import json

        for j in [job] + [j for j in self._selected() if j != job]:
            self._start(j)
        proc, out, err = self._procs[job]
        rc = proc.wait(timeout=timeout)
        err.seek(0)
        assert rc == 0, err.read()[-3000:]
        out.seek(0)
        line = [l for l in out.read().splitlines() if l.startswith("RESULT ")][-1]
        return json.loads(line[len("RESULT "):])

    def close(self):
        import shutil

        for proc, out, err in self._procs.values():
            if proc.poll() is None:
                proc.kill()  # (the exact child this fixture started)
                proc.wait()
            out.close()
            err.close()
        if self._tmp is not None:
            shutil.rmtree(self._tmp, ignore_errors=True)


_WALKS = {}


def pytest_collection_finish(session):
    """Start the walk workers of the selected walk tests right after collection: they are host-bound (CPU oracle) and finish under the other
    tests of the session instead of holding the first walk test for two minutes."""
    if getattr(session.config.option, "collectonly", False):
        return
    # Only on a box that can run them: a GPU is present (device_count() does not initialise it) and the in-tree library exists.  Elsewhere
    # (`-m "not gpu"` on the build container, a box without the .so) nothing is started; a walk test that still runs starts its own worker
    # through the fixture and fails there, next to its cause.  No test of the suite asserts a timing, so the workers' background load
    # (<= 6 processes x 32 host threads, a few seconds of GPU time each) cannot fail a neighbour.
    try:
        import torch

        if torch.cuda.device_count() < 1 or not os.path.exists(os.path.join(PKG, "libttk_hip.so")):
            return
    except Exception:  # noqa: BLE001
        return
    w = _WalkWorkers(session)
    jobs = w._selected()
    if jobs:
        for j in jobs:
            w._start(j)
        _WALKS["w"] = w


def pytest_sessionfinish(session, exitstatus):
    w = _WALKS.pop("w", None)
    if w is not None:
        w.close()


@pytest.fixture(scope="session")
def walk_workers(request):
    if "w" in _WALKS:
        yield _WALKS["w"]
        return
    w = _WalkWorkers(request.session)
    yield w
    w.close()
