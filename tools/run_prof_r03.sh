set -x
python bench.py > gpurun_out/r03_a_bench_default.log 2>&1; tail -1 gpurun_out/r03_a_bench_default.log > gpurun_out/r03_a_bench_default.json
tools/profile_gpu.sh r03_a_dmr_full > gpurun_out/r03_a_prof_dmr.log 2>&1
tools/profile_gpu.sh r03_a_ysf_full --workload ysf_full > gpurun_out/r03_a_prof_ysf.log 2>&1
tail -c 1500 gpurun_out/r03_a_bench_default.json | head -c 1500; echo; head -c 2500 gpurun_out/r03_a_bench_default.json
cat gpurun_out/prof_r03_a_dmr_full/trace_summary.txt | grep -E "k_chain|k_rrc" | cut -c1-160
