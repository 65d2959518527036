# round-2 evidence: GPU tests, bench lines, rocprofv3 kernel stats + PMC passes for the four bench workloads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/pytest_gpu.txt 2>&1; tail -2 gpurun_out/r2e/pytest_gpu.txt
python bench.py --launch-stats 300 > gpurun_out/r2e/bench_res2a_s32.json 2> gpurun_out/r2e/bench_res2a_s32.err; tail -c 600 gpurun_out/r2e/bench_res2a_s32.json
python bench.py --dst u8 --no-cpu-baseline --launch-stats 300 > gpurun_out/r2e/bench_res2a_u8.json 2>/dev/null
python bench.py --workload vgg --no-cpu-baseline --steps 50 --warmup 5 --launch-stats 100 > gpurun_out/r2e/bench_vgg_f32.json 2>/dev/null
python bench.py --workload concat --steps 100 --warmup 10 > gpurun_out/r2e/bench_concat.json 2> gpurun_out/r2e/bench_concat.err; tail -c 400 gpurun_out/r2e/bench_concat.json
echo "== pmc u8"; bash profiles/collect_pmc.sh r2_res2a_u8 --workload res2a --dst u8 > /dev/null 2>&1; head -12 gpurun_out/pmc_r2_res2a_u8/summary.txt
echo "== pmc s32"; bash profiles/collect_pmc.sh r2_res2a_s32 --workload res2a --dst s32 > /dev/null 2>&1; head -8 gpurun_out/pmc_r2_res2a_s32/summary.txt
echo "== pmc vgg"; bash profiles/collect_pmc.sh r2_vgg_f32 --workload vgg > /dev/null 2>&1; head -8 gpurun_out/pmc_r2_vgg_f32/summary.txt
echo "== pmc concat"; bash profiles/collect_pmc.sh r2_concat --workload concat > /dev/null 2>&1; head -8 gpurun_out/pmc_r2_concat/summary.txt
find gpurun_out/pmc_r2_* -name "*.csv" -size +2M -delete; find gpurun_out/pmc_r2_* -name "*.db" -delete
