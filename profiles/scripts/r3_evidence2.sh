#!/bin/bash
# Round-3 final evidence run (one box): GPU tests, driver-style default bench line, per-workload bench lines,
# rocprofv3 kernel stats + PMC passes of the headline (s32), the u8 block and the general fused blocks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r3f
python -m pytest tests -m gpu -x -q > gpurun_out/r3f/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3f/pytest_gpu.log
tail -3 gpurun_out/r3f/pytest_gpu.log
python bench.py > gpurun_out/r3f/bench_default.json 2> gpurun_out/r3f/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r3f/bench_driver_style.json 2> gpurun_out/r3f/bench_driver_style.err
python bench.py --dst u8 --steps 200 --warmup 20 --launch-stats 200 > gpurun_out/r3f/bench_res2a_u8.json 2>/dev/null
python bench.py --workload vgg --steps 100 --warmup 10 > gpurun_out/r3f/bench_vgg_f32.json 2>/dev/null
python bench.py --workload concat --steps 200 --warmup 20 > gpurun_out/r3f/bench_concat.json 2>/dev/null
for w in res3 res4 res5 res3s2; do
  python bench.py --workload $w --steps 200 --warmup 20 > gpurun_out/r3f/bench_$w.json 2>/dev/null
done
profiles/scripts/r3_general.sh > gpurun_out/r3f/general.txt 2>&1
echo "progress: bench lines done"
for t in "r3f_res4_u8 --workload res4" "r3f_res5_u8 --workload res5" "r3f_res3_u8 --workload res3" "r3f_res2a_u8 --dst u8" "r3f_res2a_s32" "r3f_vgg_f32 --workload vgg"; do
  set -- $t
  tag=$1; shift
  profiles/collect_pmc.sh $tag "$@" > /dev/null 2>&1
  echo "progress: pmc $tag done"
done
for f in gpurun_out/r3f/bench_*.json; do echo "== $f"; python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
r=d['roofline']; print(d['config'].get('kernel'), 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'mfma', r.get('mfma_frac_of_int8_peak'), 'u8_out', (d.get('u8_out') or {}).get('kernel_ms'), (d.get('u8_out') or {}).get('frac'))"; done
cat gpurun_out/r3f/general.txt
for t in r3f_res4_u8 r3f_res5_u8 r3f_res3_u8 r3f_res2a_u8 r3f_res2a_s32 r3f_vgg_f32; do echo "== $t"; grep -E "conv_|FETCH|WRITE_SIZE|SQ_INSTS_VALU |SQ_INSTS_SALU|COEXEC|MFMA_BUSY|SQ_LDS_BANK|SQ_LDS_IDX|SQ_WAIT_INST_ANY|SQ_WAVE_CYCLES" gpurun_out/pmc_$t/summary.txt | cut -c1-140; done
