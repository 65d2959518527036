#!/bin/bash
# one box, interleaved, one library: unit geometries of the u8 res2a block on the role-specialised kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for round in 1 2 3; do
  for v in "" "2,56" "3,56" "4,56" "5,56" "7,56"; do
    if [ -n "$v" ]; then export DFX_FORCE_GEOM=$v; else unset DFX_FORCE_GEOM; fi
    echo -n "res2a u8 geom=${v:-default}: "
    python bench.py --dst u8 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], d['config']['rows_per_unit'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
