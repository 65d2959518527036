#!/bin/bash
# the three general blocks with the automatic choice + stamps of the same
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
line() { python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t.strip().splitlines()[-1]); print(d['config']['kernel'][:70], 'kernel_ms', d['roofline']['kernel_ms'], 'mfma_frac', d['roofline']['mfma_frac_of_int8_peak'])
except Exception as e: print('no line', t[-300:])"; }
for w in ${WL:-res3 res4 res5}; do
  echo "== $w"; DFX_STREAM_DIRECT=1 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | line
  DFX_STREAM_DIRECT=1 timeout -k 10 100 python profiles/stamps_direct.py $w 2>&1 | grep -v amdgpu.ids
done
