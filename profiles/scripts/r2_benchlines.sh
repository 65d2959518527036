# the four bench lines of round 2 (driver-style short runs for the default workload, longer ones for the rest)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2e/bench_res2a_s32_driver_style.json 2>/dev/null
python bench.py --launch-stats 300 > gpurun_out/r2e/bench_res2a_s32.json 2>/dev/null
python bench.py --dst u8 --no-cpu-baseline --launch-stats 300 > gpurun_out/r2e/bench_res2a_u8.json 2>/dev/null
python bench.py --workload vgg --no-cpu-baseline --steps 50 --warmup 5 --launch-stats 100 > gpurun_out/r2e/bench_vgg_f32.json 2>/dev/null
python bench.py --workload concat --steps 100 --warmup 10 > gpurun_out/r2e/bench_concat.json 2>/dev/null
for f in gpurun_out/r2e/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().split('\n')[-1])
print('$f', d['steps'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('launch_ms'))"; done
