"""bench.py's launcher logic on CPU: `--gpus N` must become N ranks or fail loudly -- never a line for fewer GPUs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_present_fails_loudly():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 2 requested but this box has" in r.stderr and '"metric"' not in r.stdout


def test_world_size_must_match_gpus():
    """started by a launcher with another rank count than --gpus: refuse instead of printing n_gpus != --gpus"""
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "launch with --nproc-per-node == --gpus" in r.stderr and '"metric"' not in r.stdout


def test_effective_cores_respects_affinity():
    sys.path.insert(0, ROOT)
    import bench
    visible, eff = bench.effective_cores()
    assert 1 <= eff <= visible
    if hasattr(os, "sched_getaffinity"):
        assert eff <= len(os.sched_getaffinity(0))
        old = os.sched_getaffinity(0)
        try:
            os.sched_setaffinity(0, {min(old)})
            assert bench.effective_cores()[1] == 1
        finally:
            os.sched_setaffinity(0, old)
