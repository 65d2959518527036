#!/bin/bash
# A/B on ONE box, interleaved: role-specialised kernel vs conv_mfma.cuh's kernel (DFX_NO_ROLES=1), res2a u8 out
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
  for env in "" "DFX_NO_ROLES=1"; do
    echo "== $env" 
    env $env python bench.py --dst u8 --steps 200 --warmup 20 --no-cpu-baseline --launch-stats 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['config']['kernel'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'launch', d.get('launch_ms'))"
  done
done
