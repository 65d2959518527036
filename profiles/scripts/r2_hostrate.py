"""host enqueue rate of op.submit (python/ctypes) vs the kernel time: is the back-to-back loop host-bound?"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import cases as C, hipref
from dataclasses import replace
import bench
case, _ = bench.workloads()["res2a"]
case = replace(case, dst_dt=C.U8)
data = C.generate(case)
op = hipref.make_conv(case, data)
src = torch.from_numpy(data["src"]).cuda()
srcs = [src.clone() for _ in range(4)]
outs = [torch.empty(op.dst_shape, dtype=torch.uint8, device="cuda") for _ in range(4)]
for i in range(20): op.submit(src, outs[i % 4])
torch.cuda.synchronize()
st = torch.cuda.current_stream()
for label, kw, rot in (("default stream lookup, one src", {}, False), ("stream passed, one src", {"stream": st}, False),
                       ("stream passed, four srcs rotating", {"stream": st}, True)):
    n = 2000
    t0 = time.perf_counter()
    for i in range(n): op.submit(srcs[i % 4] if rot else src, outs[i % 4], **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host enqueue %.2f us per submit, total incl. drain %.2f us per submit" % (label, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
