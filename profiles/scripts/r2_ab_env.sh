# A/B of tuning-switch combinations of the shipped library on one box: CFGS="A=1;B=2 C=3 ..." (semicolon = several switches; values may contain commas)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload ${WLD:-res2a} --launch-stats 300"
K="import sys,json; d=json.loads(sys.stdin.read()); l=d['launch_ms']; print(d['config'].get('rows_per_unit'), d['roofline']['kernel_ms'], 'min', l['min'], 'med', l['median'], 'p90', l['p90'])"
for rep in 1 2 3; do
  for c in default $CFGS; do
    if [ "$c" = default ]; then e="X=1"; else e=$(echo $c | tr ";" " "); fi
    v=$(env $e $B --dst ${DST:-s32} 2>/dev/null | python -c "$K"); echo "$c ${DST:-s32} $v"
  done
done
