#!/bin/bash
# tools/collect_prof.sh <tag>: copy what tools/run_prof.sh <tag> left under gpurun_out/ into profiles/ (here, after the gpurun call)
TAG=${1:?tag}
[ -s gpurun_out/${TAG}_bench_default.json ] && cp gpurun_out/${TAG}_bench_default.json profiles/
[ -s gpurun_out/${TAG}_bench_default_detail.json ] && cp gpurun_out/${TAG}_bench_default_detail.json profiles/
for d in gpurun_out/prof_${TAG}_*; do
  [ -d "$d" ] || continue
  w=${d#gpurun_out/prof_${TAG}_}
  cp $d/trace_summary.txt profiles/${TAG}_${w}_kernel_trace.txt
  cat $d/pmc*_summary.txt > profiles/${TAG}_${w}_pmc.txt
  cp $d/bench_under_trace.json profiles/${TAG}_bench_${w}_under_trace.json
done
ls profiles | grep "^${TAG}_"
