#!/bin/bash
# bench lines of the general-shape workloads (streamed-weight / direct-weight kernels) + the s32 / vgg headlines
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for w in res3 res4 res5 res3s2 vgg3 vgg5 pw256 vggpool; do
  python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', d['config']['kernel'][:60], 'kernel_ms', d['roofline']['kernel_ms'], 'mfma_frac', d['roofline']['mfma_frac_of_int8_peak'], 'split', d['config']['split'])"
done
for w in "res2a" "vgg"; do
  python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-u8-out 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w', d['config']['kernel'][:60], 'kernel_ms', d['roofline']['kernel_ms'], 'hbm_frac', d['roofline']['hbm_frac_of_8TBps'])"
done
