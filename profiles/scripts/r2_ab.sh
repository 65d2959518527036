# A/B of library variants on one box: LIBS="name name ..." (deep-fusion_amd/libdfx_hip_<name>.so; "main" = shipped library)
# REPS interleaved rounds; WL = workloads to run (default "u8 s32 vgg")
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
K="import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
for rep in $(seq 1 ${REPS:-3}); do
for lib in $LIBS; do
  if [ "$lib" = main ]; then unset DFX_LIB_PATH; else export DFX_LIB_PATH=$GRAFT_REPO_ROOT/deep-fusion_amd/libdfx_hip_$lib.so; fi
  line="$lib"
  for wl in ${WL:-u8 s32 vgg}; do
    case $wl in
      u8) v=$($B --workload res2a --dst u8 2>/dev/null | python -c "$K");;
      s32) v=$($B --workload res2a --dst s32 2>/dev/null | python -c "$K");;
      vgg) v=$(python bench.py --steps 50 --warmup 5 --no-cpu-baseline --workload vgg 2>/dev/null | python -c "$K");;
    esac
    line="$line $wl $v"
  done
  echo "$line"
done
done
