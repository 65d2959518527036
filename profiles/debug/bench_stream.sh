# usage: bash profiles/debug/bench_stream.sh  -- stream-kernel workloads, one line each
run() { timeout -k 10 200 python bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(d[\"config\"][\"workload\"][:12], d[\"config\"][\"kernel\"], d[\"config\"][\"grid\"], d[\"config\"][\"lds_bytes\"], r[\"kernel_ms\"], r[\"int8_TOPs\"], r[\"hbm_GBps\"])"; }
for w in res3 res4 res5 res3s2; do run --workload $w; done
for d in s32 u8; do run --variant 3 --dst $d; done   # the headline block forced onto the general kernels
