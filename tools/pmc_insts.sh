#!/bin/bash
# Instruction counters of builds of the library on the bench workload (run on the GPU box): tools/pmc_insts.sh <proto> lib.so...
PROTO=$1; shift
export TMPDIR=/tmp
for LIB in "$@"; do
  out=gpurun_out/pmci_$(basename $LIB .so); rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-include-regex "k_chain|k_rrc_demod" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $out -o pmc -- python tools/run_lib.py $PROTO $LIB 3 > $out/log.txt 2>&1
  echo "== $PROTO $LIB"; python tools/rocpd_summary.py $out/pmc_results.db 2>&1 | grep -A200 "PMC counters" | grep -E "k_chain|k_rrc_demod" | cut -c60-130
  rm -f $out/*.db
done
