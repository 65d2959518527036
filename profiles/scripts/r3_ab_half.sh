#!/bin/bash
# A/B on one box, interleaved, one library: the last units of a store-bound op handed out as half units (default)
# against whole units (DFX_HALF_UNITS=0) and against more of them
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
one() {  # label, env value or "", bench args
  echo -n "$1: "; if [ -n "$2" ]; then export DFX_HALF_UNITS=$2; else unset DFX_HALF_UNITS; fi
  python bench.py ${@:3} --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
}
for round in 1 2 3; do
  for v in 0 "" 256 1024 2048; do
    one "res2a s32 half=${v:-default}" "$v" --steps 300 --warmup 30 --no-u8-out
  done
done
for round in 1 2; do
  for v in 0 "" 2048; do
    one "vgg f32 half=${v:-default}" "$v" --workload vgg --steps 100 --warmup 10
  done
done
