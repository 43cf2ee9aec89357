#!/bin/bash
# tools/run_prof.sh <tag> [workloads...]: what produces profiles/<tag>_* (run on the GPU box through gpurun).
#   default bench line (all other_configs), then for every workload (default: dmr_full ysf_full nxdn_full rrc_gfsk) a kernel trace
#   and separate PMC passes (tools/profile_gpu.sh); rrc_gfsk runs at 4 096 channels, the size BASELINE configs[1] names.
#   SKIP_BENCH=1 leaves the default bench line out.  Afterwards: tools/collect_prof.sh <tag> copies the summaries into profiles/.
TAG=${1:?tag}; shift
WL=${*:-dmr_full ysf_full nxdn_full rrc_gfsk rrc_gfsk_one}
set -x
mkdir -p gpurun_out
if [ -z "${SKIP_BENCH:-}" ]; then
  python bench.py > gpurun_out/${TAG}_bench_default.log 2>&1; tail -1 gpurun_out/${TAG}_bench_default.log > gpurun_out/${TAG}_bench_default.json
  grep '^BENCH_DETAIL ' gpurun_out/${TAG}_bench_default.log | tail -1 | cut -d' ' -f2- > gpurun_out/${TAG}_bench_default_detail.json
  wc -c gpurun_out/${TAG}_bench_default.json
  head -c 2500 gpurun_out/${TAG}_bench_default.json; echo
fi
for w in $WL; do
  extra=""; case "$w" in rrc_gfsk*) extra="--channels 4096";; esac
  tools/profile_gpu.sh ${TAG}_$w --workload $w $extra > gpurun_out/${TAG}_prof_$w.log 2>&1
  grep -E "k_chain|k_rrc" gpurun_out/prof_${TAG}_$w/trace_summary.txt | cut -c1-160
done
