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


def _full_record():
    """A full bench record as bench.py builds it before compacting: round 5's own 20.7 KB line (the one the driver could not parse)."""
    import json
    return json.loads(open(os.path.join(ROOT, "profiles", "r05_j_bench_default.json")).read().strip().splitlines()[-1])


def test_final_line_is_small_enough_for_the_driver_to_parse():
    """BENCH_r05.parsed was null: the last stdout line had grown to 20.7 KB and the driver keeps an 8 KB tail.  The final line
    is now a compact record, hard-bounded at 4 KB, that still carries the headline, `roofline` and `cpu_baseline`."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = _full_record()
    assert len(json.dumps(full)) > 8192                         # (the fixture really is the oversized one)
    text = bench.compact_line(full, "bench_detail.json")
    assert len(text) <= 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["value"] == float("%.6g" % full["value"]) and line["ms_per_step"] == float("%.6g" % full["ms_per_step"])
    assert line["config"]["channels_per_gpu"] == 16384 and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["frac"] - full["roofline"]["frac"]) < 1e-5 and r["traffic"] and r["avg_launch_ms"] > 0
    assert set(r["co_limit"]["issue"]) >= {"vector", "scalar", "lds", "mfma", "branch", "simd_cycles_per_run"}
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["gpu_matches_baseline_outputs"] is True
    assert line["verified"]["bit_exact_vs_oracle"] is True
    oc = line["other_configs"]
    assert len(oc) == len(full["other_configs"])
    for row, e in zip(oc, full["other_configs"]):
        assert row["workload"] == e["workload"] and row["ok"] is True and abs(row["frac"] - e["frac"]) < 1e-5
        assert len(json.dumps(row)) < 220


def test_final_line_stays_bounded_whatever_the_record_holds():
    """Three times as many workloads with long names, prose where numbers should be: the headline, roofline.frac and
    cpu_baseline.value survive, the least important parts are replaced by a pointer to the detail file."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = _full_record()
    full["other_configs"] = [dict(e, workload=e["workload"] + " " + "x" * 200) for e in full["other_configs"] * 3]
    full["config"]["workload"] = "w" * 5000
    text = bench.compact_line(full, "bench_detail.json")
    assert len(text) <= 4096
    line = json.loads(text)
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["value"] > 0
    assert line["other_configs"] == "see bench_detail.json"
    # a failed other_configs run is reported as such, not dropped
    full = _full_record()
    full["other_configs"] = {"error": "RuntimeError('x')"}
    assert json.loads(bench.compact_line(full))["other_configs"] == {"error": "RuntimeError('x')"}
