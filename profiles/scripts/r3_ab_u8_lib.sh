#!/bin/bash
# A/B on one box, interleaved: the shipped library against variant builds on the u8 res2a block
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for round in 1 2 3 4; do
  for v in "" ${VARIANTS}; do
    lib=$R/deep-fusion_amd/libdfx_hip${v:+_$v}.so
    echo -n "res2a u8 ${v:-shipped}: "; DFX_LIB_PATH=$lib python bench.py --dst u8 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], d['roofline']['kernel_ms'])"
  done
done
