#!/bin/bash
# Round-3 evidence run (one box): GPU tests, driver-style default bench line, per-workload bench lines,
# rocprofv3 kernel stats + PMC passes of the default (s32) and the u8 workloads.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r3e
python -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3e/pytest_gpu.log
tail -3 gpurun_out/r3e/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3e/bench_driver_style.json 2> gpurun_out/r3e/bench_driver_style.err
python bench.py --steps 200 --warmup 20 --launch-stats 200 > gpurun_out/r3e/bench_res2a_s32.json 2>/dev/null
python bench.py --dst u8 --steps 200 --warmup 20 --launch-stats 200 > gpurun_out/r3e/bench_res2a_u8.json 2>/dev/null
python bench.py --workload vgg --steps 100 --warmup 10 > gpurun_out/r3e/bench_vgg_f32.json 2>/dev/null
python bench.py --workload concat --steps 200 --warmup 20 > gpurun_out/r3e/bench_concat.json 2>/dev/null
profiles/scripts/r3_general.sh > gpurun_out/r3e/general.txt 2>&1
profiles/collect_pmc.sh r3_res2a_u8 --dst u8 > /dev/null 2>&1
profiles/collect_pmc.sh r3_res2a_s32 > /dev/null 2>&1
profiles/collect_pmc.sh r3_vgg_f32 --workload vgg > /dev/null 2>&1
for f in gpurun_out/r3e/bench_*.json; do echo "== $f"; python3 -c "
import json,sys
d=json.load(open('$f'))
r=d['roofline']; print(d['config'].get('kernel'), 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'u8_out', (d.get('u8_out') or {}).get('kernel_ms'), (d.get('u8_out') or {}).get('frac'))"; done
cat gpurun_out/r3e/general.txt
for t in r3_res2a_u8 r3_res2a_s32 r3_vgg_f32; do echo "== $t"; grep -E "conv_|FETCH|WRITE_SIZE|SQ_INSTS_VALU |SQ_INSTS_SALU|COEXEC|MFMA_BUSY|SQ_LDS_BANK|SQ_LDS_IDX|SQ_WAIT_INST_ANY|SQ_WAVE_CYCLES" gpurun_out/pmc_$t/summary.txt | cut -c1-140; done
