#!/bin/bash
# general fused blocks on conv_direct.cuh by waves per workgroup (DFX_DIRECT_NW), conv1 split (DFX_DIRECT_WO1) and
# unit size (DFX_DIRECT_NPB), against the automatic choice
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
line() { python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t.strip().splitlines()[-1]); print(d['config']['kernel'][:70], 'kernel_ms', d['roofline']['kernel_ms'], 'mfma_frac', d['roofline']['mfma_frac_of_int8_peak'])
except Exception as e: print('no line', t[-300:])"; }
WL=${WL:-"res3 res4 res5 res3s2"}
for w in $WL; do
  echo "== $w auto"; python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | line
  for cfg in "4 1 4" "4 4 4" "4 4 2" "4 4 1" "8 8 4" "8 8 2" "8 8 1"; do
    set -- $cfg
    echo "== $w direct nw $1 wo1 $2 npb $3"; DFX_STREAM_DIRECT=1 DFX_DIRECT_NW=$1 DFX_DIRECT_WO1=$2 DFX_DIRECT_NPB=$3 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | line
  done
done
