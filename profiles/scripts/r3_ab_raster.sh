#!/bin/bash
# raster units against rectangular patches (DFX_DIRECT_RASTER=1/0) on one box, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for round in 1 2 3; do
for w in ${WL:-res3 res4 res5 res3s2}; do
  for r in 0 1; do
    echo -n "$w raster=$r: "; DFX_DIRECT_RASTER=$r python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], 'lds', d['config']['lds_bytes'], d['roofline']['kernel_ms'], d['roofline']['mfma_frac_of_int8_peak'])"
  done
done
done
