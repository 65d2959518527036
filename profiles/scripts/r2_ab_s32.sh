cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload res2a --launch-stats 300"
K="import sys,json; d=json.loads(sys.stdin.read()); l=d['launch_ms']; print(d['roofline']['kernel_ms'], 'min', l['min'], 'med', l['median'], 'p90', l['p90'])"
run() { # name lib env dst
  if [ "$2" = main ]; then unset DFX_LIB_PATH; else export DFX_LIB_PATH=$GRAFT_REPO_ROOT/deep-fusion_amd/libdfx_hip_$2.so; fi
  v=$(env $3 $B --dst $4 2>/dev/null | python -c "$K"); echo "$1 $4 $v"
}
for rep in 1 2 3; do
  for dst in ${DSTS:-s32}; do
    for v in ${VARS:-main early c349}; do run $v $v X=1 $dst; done
    for e in $ENVS; do run main_$e main $e $dst; done
  done
done
