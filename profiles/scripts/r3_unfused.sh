#!/bin/bash
# unfused general convs: conv_direct.cuh (default where it fits) against conv_stream.cuh (DFX_STREAM_DIRECT=0), one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for round in 1 2; do
for w in vgg3 vgg5 pw256 pw1024; do
  for dflag in 1 0; do
    echo -n "$w direct=$dflag: "; DFX_STREAM_DIRECT=$dflag python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], 'lds', d['config']['lds_bytes'], d['roofline']['kernel_ms'], d['roofline']['mfma_frac_of_int8_peak'])"
  done
done
done
