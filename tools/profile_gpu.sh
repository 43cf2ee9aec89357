#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of the headline bench, summaries into gpurun_out/prof_$1
#   tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --verify 0 $*"
# (the trace pass runs bench.py's own default step counts -- 2 warm-up + 10 timed launches -- so that its per-kernel average is over the same
# launches the bench line's HIP events time; the first launches of a process are slower and weighed a 4-launch average down: r06_d 5.57 traced vs 5.04 live)
TRACE_ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --verify 0 $*"
if [ -z "${SKIP_TRACE:-}" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py $TRACE_ARGS > $OUT/trace.log 2>&1; fi
grep '^{' $OUT/trace.log > $OUT/bench_under_trace.json
python tools/rocpd_summary.py $OUT/trace_results.db > $OUT/trace_summary.txt 2>&1
# PMC passes: counters only, no tracing domains (gpurun refuses --pmc together with sys/hip/hsa traces)
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-include-regex "k_chain|k_rrc_demod|k_dmr|k_ysf|k_rrc_tile|k_nxdn" --pmc $set -d $OUT -o pmc$i -- python bench.py $ARGS > $OUT/pmc$i.log 2>&1
  # what the pass ran on, from its own bench line: bench.py scales the counters by it (profiled_counters)
  CFG=$(grep '^{' $OUT/pmc$i.log | tail -1 | python -c 'import json,sys; c=json.loads(sys.stdin.read())["config"]; print("channels_per_gpu=%d samples_per_channel=%d" % (c["channels_per_gpu"], c["samples_per_channel_per_step"]))' 2>/dev/null)
  MS=$(grep '^BENCH_DETAIL ' $OUT/pmc$i.log | tail -1 | cut -d' ' -f2- | python -c 'import json,sys; print("avg_launch_ms=%.6f" % json.loads(sys.stdin.read())["roofline"]["avg_launch_ms"])' 2>/dev/null)
  CFG="$CFG $MS"
  { echo "## config: $CFG args=\"$ARGS\""; python tools/rocpd_summary.py $OUT/pmc${i}_results.db 2>&1 | grep -A200 "PMC counters" | grep -E "k_chain|k_rrc_demod|k_dmr|k_ysf|k_rrc_tile|PMC"; } > $OUT/pmc${i}_summary.txt
done
rm -f $OUT/*.db
cat $OUT/trace_summary.txt | grep -v "at::native\|rocclr\|rtc\|twiddle\|r2c\|c2r" | head -12 | cut -c1-170
cat $OUT/pmc*_summary.txt | cut -c1-200
