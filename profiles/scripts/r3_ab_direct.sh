#!/bin/bash
# A/B on one box: the shipped library against variant builds (make variant NAME=...), general blocks, 3 rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
line() { python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t.strip().splitlines()[-1]); print(d['config']['kernel'][:60], 'kernel_ms', d['roofline']['kernel_ms'], 'mfma_frac', d['roofline']['mfma_frac_of_int8_peak'])
except Exception as e: print('no line', t[-300:])"; }
for round in 1 2 3; do
for w in ${WL:-res3 res4 res5}; do
  for v in "" ${VARIANTS}; do
    lib=$R/deep-fusion_amd/libdfx_hip${v:+_$v}.so
    echo -n "$w ${v:-shipped}: "; DFX_LIB_PATH=$lib DFX_STREAM_DIRECT=1 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | line
  done
done
done
