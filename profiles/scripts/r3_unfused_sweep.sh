#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for w in vgg3 vgg5; do
  for cfg in "4 1 4" "4 4 2" "4 4 1" "8 8 4" "8 8 2" "8 8 1"; do
    set -- $cfg
    echo -n "$w nw $1 wo1 $2 npb $3: "; DFX_DIRECT_NW=$1 DFX_DIRECT_WO1=$2 DFX_DIRECT_NPB=$3 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], 'grid', d['config']['grid'], 'lds', d['config']['lds_bytes'], d['roofline']['kernel_ms'], d['roofline']['mfma_frac_of_int8_peak'])"
  done
done
