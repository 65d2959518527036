#!/bin/bash
# A/B on one box, interleaved: libdfx_hip_base.so (the previous build) against the shipped library on the three
# resident-weight headline kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
one() {  # label, lib, bench args
  echo -n "$1: "; DFX_LIB_PATH=$2 python bench.py ${@:3} --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
}
for round in 1 2 3 4; do
  for v in _base ""; do
    lib=$R/deep-fusion_amd/libdfx_hip$v.so
    one "res2a u8 ${v:-_new}" $lib --dst u8 --steps 300 --warmup 30
    one "res2a s32 ${v:-_new}" $lib --steps 300 --warmup 30 --no-u8-out
  done
done
for round in 1 2; do
  for v in _base ""; do
    lib=$R/deep-fusion_amd/libdfx_hip$v.so
    one "vgg f32 ${v:-_new}" $lib --workload vgg --steps 100 --warmup 10
    one "res2a s8 ${v:-_new}" $lib --dst s8 --steps 300 --warmup 30
  done
done
