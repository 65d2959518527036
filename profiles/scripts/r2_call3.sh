cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r2/pytest_gpu_full.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r2/pytest_gpu_full.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload res2a"
for cfg in "" "DFX_FORCE_GEOM=2,56" "DFX_STATIC_ROUNDS=0" "DFX_NO_MAGIC=1"; do
  echo "== u8 $cfg"; env $cfg $B --dst u8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['rows_per_unit'], d['roofline']['mfma_frac_of_int8_peak'])"
done
for cfg in "" "DFX_FORCE_GEOM=1,56" "DFX_FORCE_GEOM=4,56" "DFX_STATIC_ROUNDS=0"; do
  echo "== s32 $cfg"; env $cfg $B --dst s32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['rows_per_unit'], d['roofline']['frac'])"
done
echo "== vgg f32"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload vgg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['rows_per_unit'], d['roofline']['frac'])"
