# one line per resident-kernel workload (kernel ms by HIP events)
run() { timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(d[\"config\"][\"workload\"][:10], d[\"config\"][\"kernel\"], r[\"kernel_ms\"], r[\"hbm_GBps\"], r[\"int8_TOPs\"])"; }
run; run; run --dst u8; run --dst f32; run --workload vgg; run --workload bringup
