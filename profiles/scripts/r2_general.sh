# general-shape kernels: default choice vs the tuning switches (one box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
K="import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel'][:60], d['roofline']['kernel_ms'], d['roofline']['frac'])"
for w in res3 res4 res5 res3s2; do
  for e in X=1 DFX_STREAM_PXB=2 DFX_STREAM_DIRECT=1 DFX_STREAM_DIRECT=0 DFX_STREAM_SPLIT=0 DFX_STREAM_SPLIT=1; do
    v=$(env $e python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload $w 2>/dev/null | python -c "$K" 2>/dev/null); echo "$w $e $v"
  done
done
