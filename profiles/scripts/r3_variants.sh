#!/bin/bash
# res2a u8 on the product build and on diagnostic variants of the library (libdfx_hip_<name>.so), one box, interleaved
#   usage: r3_variants.sh <name...>      ("main" = the product build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  for v in "$@"; do
    lib=$R/deep-fusion_amd/libdfx_hip_$v.so
    [ "$v" = main ] && lib=$R/deep-fusion_amd/libdfx_hip.so
    echo -n "== $v: "
    DFX_LIB_PATH=$lib python bench.py --dst u8 --steps 200 --warmup 20 --no-cpu-baseline --launch-stats 100 2>&1 | tail -1 | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t)
    print(d['config']['kernel'], 'kernel_ms', d['roofline']['kernel_ms'], 'launch', d.get('launch_ms'))
except Exception as e:
    print('no line:', t[-300:])"
  done
done
