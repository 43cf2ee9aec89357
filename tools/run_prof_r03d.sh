# round 3, after the tail split of the chain launches: default bench line, kernel traces + PMC passes of the DMR / YSF / NXDN chains and the config-1 slicer
set -x
python bench.py > gpurun_out/r03_d_bench_default.log 2>&1; tail -1 gpurun_out/r03_d_bench_default.log > gpurun_out/r03_d_bench_default.json
tools/profile_gpu.sh r03_d_dmr_full > gpurun_out/r03_d_prof_dmr.log 2>&1
tools/profile_gpu.sh r03_d_ysf_full --workload ysf_full > gpurun_out/r03_d_prof_ysf.log 2>&1
tools/profile_gpu.sh r03_d_nxdn_full --workload nxdn_full > gpurun_out/r03_d_prof_nxdn.log 2>&1
tools/profile_gpu.sh r03_d_rrc_gfsk --workload rrc_gfsk > gpurun_out/r03_d_prof_rrc.log 2>&1
python tools/push_size.py > gpurun_out/r03_d_push_size.txt 2>&1
head -c 3000 gpurun_out/r03_d_bench_default.json; echo
for w in dmr_full ysf_full nxdn_full rrc_gfsk; do grep -E "k_chain|k_rrc" gpurun_out/prof_r03_d_$w/trace_summary.txt | cut -c1-160; done
cat gpurun_out/r03_d_push_size.txt
