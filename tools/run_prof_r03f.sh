# final state of round 3 (tail split, branch weights, compile-time slicer levels): default bench line, kernel traces + PMC passes of the DMR / YSF / NXDN chains and the config-1 slicer
set -x
python bench.py > gpurun_out/r03_f_bench_default.log 2>&1; tail -1 gpurun_out/r03_f_bench_default.log > gpurun_out/r03_f_bench_default.json
tools/profile_gpu.sh r03_f_dmr_full > gpurun_out/r03_f_prof_dmr.log 2>&1
tools/profile_gpu.sh r03_f_ysf_full --workload ysf_full > gpurun_out/r03_f_prof_ysf.log 2>&1
tools/profile_gpu.sh r03_f_nxdn_full --workload nxdn_full > gpurun_out/r03_f_prof_nxdn.log 2>&1
tools/profile_gpu.sh r03_f_rrc_gfsk --workload rrc_gfsk > gpurun_out/r03_f_prof_rrc.log 2>&1
python tools/push_size.py > gpurun_out/r03_f_push_size.txt 2>&1
head -c 3000 gpurun_out/r03_f_bench_default.json; echo
for w in dmr_full ysf_full nxdn_full rrc_gfsk; do grep -E "k_chain|k_rrc" gpurun_out/prof_r03_f_$w/trace_summary.txt | cut -c1-160; done
cat gpurun_out/r03_f_push_size.txt
