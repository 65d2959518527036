cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload res2a"
for cfg in "" "DFX_FORCE_GEOM=2,56" "DFX_FORCE_GEOM=1,56" "DFX_FORCE_GEOM=2,56 DFX_STATIC_ROUNDS=6" "DFX_FORCE_GEOM=4,56 DFX_STATIC_ROUNDS=0"; do
  echo "== u8 $cfg"; env $cfg $B --dst u8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['rows_per_unit'])"
done
for cfg in "" "DFX_FORCE_GEOM=1,56" "DFX_STATIC_ROUNDS=0" "DFX_STATIC_ROUNDS=5"; do
  echo "== s32 $cfg"; env $cfg $B --dst s32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['rows_per_unit'])"
done
