#!/bin/bash
# A/B on ONE box, interleaved: role-specialised kernel vs conv_mfma.cuh's kernel (DFX_NO_ROLES=1), res2a u8 out;
# plus the s32 headline.  usage: r3_ab_u8.sh [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
REPS=${1:-3}
line() {
  python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['config']['kernel'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'launch', d.get('launch_ms'))"
}
for rep in $(seq $REPS); do
  for env in "DFX_X=0" "DFX_NO_ROLES=1"; do
    echo "== u8 $env"
    env $env python bench.py --dst u8 --steps 200 --warmup 20 --no-cpu-baseline --launch-stats 200 2>/dev/null | line
  done
  echo "== s32"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-u8-out --launch-stats 200 2>/dev/null | line
done
