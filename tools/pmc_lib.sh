#!/bin/bash
# Counter passes of one build of the library on the bench workload (run on the GPU box):
#   tools/pmc_lib.sh <proto> <lib.so> <tag>   ->  gpurun_out/pmc_<tag>/summary.txt
PROTO=$1; LIB=$2; TAG=$3
export TMPDIR=/tmp
out=gpurun_out/pmc_$TAG; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-include-regex "k_chain|k_rrc_demod" --pmc $set -d $out -o pmc$i -- python tools/run_lib.py $PROTO $LIB 3 > $out/log$i.txt 2>&1
  python tools/rocpd_summary.py $out/pmc${i}_results.db 2>&1 | grep -A200 "PMC counters" | grep -E "k_chain|k_rrc_demod|PMC" >> $out/summary.txt
done
rm -f $out/*.db
cut -c1-220 $out/summary.txt
