#!/usr/bin/env python3
"""tools/show_bench.py <bench line .json>: the numbers of a bench.py line at a glance (headline, ceiling, other_configs, checks)."""
import json
import sys

# the full record: bench_detail.json, or the `BENCH_DETAIL ` line of a bench log (the last stdout line is the compact one)
text = open(sys.argv[1]).read().strip()
det = [l for l in text.splitlines() if l.startswith("BENCH_DETAIL ")]
d = json.loads(det[-1][len("BENCH_DETAIL "):]) if det else json.loads(text) if text.startswith("{\n") else json.loads(text.splitlines()[-1])
r = d["roofline"]
print("%s: %.3f ms/step, %.2f M channels, kernel %.3f ms, frac %.3f (of achievable %.3f, ceiling %.0f GB/s %s), traffic/alg %s"
      % (d["config"]["workload"][:40], d["ms_per_step"], d["value"] / 1e6, r["avg_launch_ms"], r["frac"], r["frac_of_achievable"], r["peak_achievable"],
         {k: round(v) for k, v in (r.get("peak_achievable_rates") or {}).items()},
         "%.3f" % (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else None))
print("verified:", d.get("verified"))
cb = d.get("cpu_baseline") or {}
print("cpu_baseline:", {k: cb.get(k) for k in ("value", "cores", "kind", "msamples_per_s", "gpu_matches_baseline_outputs", "channels_hashed", "error") if k in cb})
oc = d.get("other_configs")
if isinstance(oc, dict):
    print("other_configs:", oc)
for o in oc or []:
    if not isinstance(o, dict):
        continue
    v = o.get("verified") or {}
    print("  %-28s %-22s %7.3f ms  kernel %7.3f ms  frac %.3f  traffic/alg %-6s  ok=%s%s"
          % (o["workload"], o["config"].split(" x ")[0], o["ms_per_step"], o["avg_launch_ms"], o["frac"],
             "%.3f" % (o["traffic"] / (o.get("algorithmic_bytes_per_step") or o["algorithmic_bytes_per_launch"])) if o.get("traffic") else "-",
             v.get("bit_exact_vs_oracle") or (bool(v.get("floats_within_2.5e-6_vs_oracle", v.get("within_1e-6_vs_oracle"))) and v.get("dibits_bit_exact_vs_oracle", True)),
             " (floats: %.2e%s)" % (v["max_rel_err"], ", dibits exact" if v.get("dibits_bit_exact_vs_oracle") else "") if "max_rel_err" in v else ""))
