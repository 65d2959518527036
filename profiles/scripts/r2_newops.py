"""times the pooling and eltwise-sum kernels at VGG / ResNet sizes (HIP events, rotating buffers)"""
import importlib, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
dfa = importlib.import_module("deep-fusion_amd")
def timeit(fn, n=100, warm=10):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
# conv1_2 of VGG at N=64: 224x224x64 u8 -> 112x112x64 (the pooling stage behind BASELINE configs[5]'s conv)
for (bs, h, w, c) in ((64, 224, 224, 64), (128, 56, 56, 256)):
    nb = 5
    srcs = [torch.randint(0, 255, (bs, h, w, c), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    dsts = [torch.empty((bs, h // 2, w // 2, c), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    op = dfa.Pool(bs, c, h, w, h // 2, w // 2, (2, 2), (2, 2), (0, 0), np.uint8)
    ms = timeit(lambda i: op.submit(srcs[i % nb], dsts[i % nb]))
    by = srcs[0].numel() + dsts[0].numel()
    out.append({"op": "maxpool 2x2/2 u8", "shape": [bs, h, w, c], "ms": round(ms, 5), "GBps": round(by / ms / 1e6, 1), "algorithmic_bytes": by})
for (bs, h, w, c) in ((128, 56, 56, 256),):
    nb = 4
    a = [torch.randint(0, 127, (bs, h, w, c), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    b = [torch.randint(0, 127, (bs, h, w, c), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    d = [torch.empty((bs, h, w, c), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    op = dfa.EltwiseSum(2, a[0].numel(), np.uint8, True)
    ms = timeit(lambda i: op.submit([a[i % nb], b[i % nb]], d[i % nb]))
    by = 3 * a[0].numel()
    out.append({"op": "eltwise sum+relu u8, 2 inputs", "shape": [bs, h, w, c], "ms": round(ms, 5), "GBps": round(by / ms / 1e6, 1), "algorithmic_bytes": by})
for o in out: print(json.dumps(o))
