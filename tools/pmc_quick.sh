#!/bin/bash
# tools/pmc_quick.sh <tag> [bench args]: instruction-count + clock PMC passes of the headline bench (no trace)
TAG=${1:-q}; shift || true
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --verify 0 $*"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-include-regex "k_chain|k_rrc_demod|k_dmr|k_ysf|k_rrc_tile" --pmc $set -d $OUT -o pmc$i -- python bench.py $ARGS > $OUT/pmc$i.log 2>&1
  python tools/rocpd_summary.py $OUT/pmc${i}_results.db 2>&1 | grep -A200 "PMC counters" | grep -E "k_chain|k_rrc_demod|k_dmr|k_ysf|k_rrc_tile|PMC" > $OUT/pmc${i}_summary.txt
  grep '^{' $OUT/pmc$i.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['stage_ms'])"
done
rm -f $OUT/*.db
cat $OUT/pmc*_summary.txt | cut -c1-200
