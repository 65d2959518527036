#!/bin/bash
# A/B on one box, interleaved: libdfx_hip_base.so (the previous build) against the shipped library on the general fused blocks
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for round in 1 2 3; do
  for w in res3 res4 res5 res3s2 vgg3; do
    for v in _base ""; do
      lib=$R/deep-fusion_amd/libdfx_hip$v.so
      echo -n "$w ${v:-_new}: "; DFX_LIB_PATH=$lib python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
    done
  done
done
