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


def test_profiled_counters_are_scaled_to_the_size_of_the_run():
    """The committed PMC passes say what they ran on (a `## config:` line, or the bench line under the tracer beside them);
    bench.py scales their per-launch bytes and instruction counts to the size it reports on -- never a silent assumption
    (round 3 reported a 16 384-channel pass against a 4 096-channel launch: "4.1 x wasted traffic" that was a unit bug)."""
    import bench
    full = bench.profiled_counters("rrc_gfsk", 16384, 190080)
    quarter = bench.profiled_counters("rrc_gfsk", 4096, 190080)
    assert full and quarter and ("scaled" in quarter["traffic_source"]) != ("scaled" in full["traffic_source"])      # (one of the two sizes is the one the pass ran on)
    assert abs(quarter["traffic"] / full["traffic"] - 0.25) < 1e-9
    alg = 4096 * 190080 * 8.0                       # 4 B in + 4 B out per sample
    assert 0.9 < quarter["traffic"] / alg < 1.3     # HBM traffic of the materialised-RRC kernel is its algorithmic bytes, give or take
    dmr = bench.profiled_counters("dmr_full", 16384, 190080)
    assert dmr and 0.9 < dmr["traffic"] / (16384 * 190080 * 4.108) < 1.3
