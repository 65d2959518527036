cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload res2a"
for cfg in "" $EXTRA_U8; do
  echo "== u8 $cfg"; env $cfg $B --dst u8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['rows_per_unit'], d['roofline']['mfma_frac_of_int8_peak'])"
done
for cfg in "" $EXTRA_S32; do
  echo "== s32 $cfg"; env $cfg $B --dst s32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['rows_per_unit'], d['roofline']['frac'])"
done
echo "== vgg f32"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload vgg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['config']['rows_per_unit'], d['roofline']['frac'])"
if [ -n "$STAMPS" ]; then
timeout -k 5 100 python profiles/stamps.py u8 2>&1 | grep -v amdgpu.ids
timeout -k 5 100 python profiles/stamps.py s32 2>&1 | grep -v amdgpu.ids
fi
