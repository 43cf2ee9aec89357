#!/bin/bash
# PMC passes over tools/fir_bench.py (default library): per-kernel instruction / wait counters
OUT=gpurun_out/prof_${1:-firbench}
mkdir -p $OUT; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-include-regex "k_rrc_demod|k_rrc_tile" --pmc $set -d $OUT -o pmc$i -- python tools/fir_bench.py > $OUT/pmc$i.log 2>&1
  python tools/rocpd_summary.py $OUT/pmc${i}_results.db 2>&1 | grep -A200 "PMC counters" | grep -v "^## PMC" > $OUT/pmc${i}_summary.txt
done
rm -f $OUT/*.db
cat $OUT/pmc*_summary.txt | cut -c1-170
