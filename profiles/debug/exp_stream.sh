# timing-by-elimination of conv_stream.cuh (make -C deep-fusion_amd/csrc exp EXP=n builds libdfx_hip_expN.so;
# results of those builds are wrong by design, only their run time is read)
run() { timeout -k 10 200 python bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(d[\"config\"][\"workload\"][:8], d[\"config\"][\"kernel\"], d[\"config\"][\"grid\"], r[\"kernel_ms\"])"; }
names=("baseline" "no weight loads" "no epilogue" "no MFMA" "no fragment LDS reads" "no step barrier" "no tile loads")
for e in 0 1 2 3 4 5 6; do
  if [ $e = 0 ]; then unset DFX_LIB_PATH; else export DFX_LIB_PATH=$PWD/deep-fusion_amd/libdfx_hip_exp$e.so; fi
  echo "== EXP $e: ${names[$e]}"
  for w in "$@"; do run --workload $w --variant 3; done
done
