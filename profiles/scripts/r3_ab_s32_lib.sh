#!/bin/bash
# A/B on one box, interleaved: the shipped library against variant builds on the s32 headline and vgg f32
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for round in 1 2 3 4; do
  for v in "" ${VARIANTS}; do
    lib=$R/deep-fusion_amd/libdfx_hip${v:+_$v}.so
    echo -n "res2a s32 ${v:-shipped}: "; DFX_LIB_PATH=$lib python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-u8-out 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
for round in 1 2; do
  for v in "" ${VARIANTS}; do
    lib=$R/deep-fusion_amd/libdfx_hip${v:+_$v}.so
    echo -n "vgg f32 ${v:-shipped}: "; DFX_LIB_PATH=$lib python bench.py --workload vgg --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
