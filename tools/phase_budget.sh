#!/bin/bash
# Per-phase instruction budget of the chain kernel (run on the GPU box): builds made with -DDH_STOP_AFTER=n /
# -DDH_SKIP_DECODER (tools/build_variant.sh) are run under a counter pass each; the differences between consecutive
# builds are the phases' shares.   tools/phase_budget.sh <proto> variants/lib_a.so variants/lib_b.so ...
PROTO=$1; shift
export TMPDIR=/tmp
for lib in "$@"; do
  out=gpurun_out/budget_$(basename $lib .so); rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-include-regex "k_chain|k_rrc_demod" --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU} -d $out -o pmc -- python tools/run_lib.py $PROTO $lib 3 > $out/log.txt 2>&1
  echo "== $(basename $lib)"
  python tools/rocpd_summary.py $out/pmc_results.db 2>&1 | grep -E "k_chain|k_rrc_demod" | awk '{print $(NF-3), $(NF-1)}' | tr '\n' ' ' | tee $out/summary.txt; echo
  rm -f $out/*.db
done
