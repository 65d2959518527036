#!/bin/bash
# Timing experiments (never product numbers): res2a u8 on diagnostic variants of the library, same box, interleaved.
#   usage: r3_experiments.sh <variant names...>   (libdfx_hip_<name>.so; "main" = the product build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
line() {
  python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t)
    print(d['config']['kernel'], 'kernel_ms', d['roofline']['kernel_ms'], 'launch', d.get('launch_ms'))
except Exception as e:
    print('no line:', t[-300:])"
}
for rep in 1 2; do
  for v in "$@"; do
    lib=$R/deep-fusion_amd/libdfx_hip_$v.so
    [ "$v" = main ] && lib=$R/deep-fusion_amd/libdfx_hip.so
    for roles in 0 1; do
      echo "== $v DFX_NO_ROLES=$roles"
      DFX_LIB_PATH=$lib DFX_NO_ROLES=$roles python bench.py --dst u8 --steps 200 --warmup 20 --no-cpu-baseline --launch-stats 100 2>&1 | tail -1 | line
    done
  done
done
