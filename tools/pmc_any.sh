#!/bin/bash
# Any counters of builds of the library on the bench workload (run on the GPU box): tools/pmc_any.sh <proto> "<counters>" lib.so...
PROTO=$1; CTRS=$2; shift; shift
export TMPDIR=/tmp
for LIB in "$@"; do
  out=gpurun_out/pmca_$(basename $LIB .so); rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-include-regex "k_chain|k_rrc_demod" --pmc $CTRS -d $out -o pmc -- python tools/run_lib.py $PROTO $LIB 3 > $out/log.txt 2>&1
  echo "== $PROTO $LIB"; python tools/rocpd_summary.py $out/pmc_results.db 2>&1 | grep -A200 "PMC counters" | grep -E "k_chain|k_rrc_demod" | grep -v "10, 1>" | cut -c60-140
  rm -f $out/*.db
done
