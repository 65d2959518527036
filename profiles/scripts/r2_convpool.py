"""conv + relu + 2x2/2 max pool as two launches on one stream: is the intermediate served from cache?
times conv alone, pool alone (input rotating: cold) and the pair (pool reads what the conv just wrote)"""
import importlib, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import cases as C, hipref
dfa = importlib.import_module("deep-fusion_amd")
def timeit(fn, n=60, warm=10):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (bs, hw, ic, oc) in ((64, 224, 64, 64), (128, 56, 64, 64)):
    case = C.ConvCase("cp", bs, ic, hw, hw, oc, 0, dst_dt=C.U8)
    data = C.generate(case)
    op = hipref.make_conv(case, data)
    nb = 3
    srcs = [torch.from_numpy(np.roll(data["src"], i, axis=0)).cuda() for i in range(nb)]
    mids = [torch.empty(op.dst_shape, dtype=torch.uint8, device="cuda") for _ in range(nb)]
    outs = [torch.empty((bs, hw // 2, hw // 2, oc), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    pool = dfa.Pool(bs, oc, hw, hw, hw // 2, hw // 2, (2, 2), (2, 2), (0, 0), np.uint8)
    t_conv = timeit(lambda i: op.submit(srcs[i % nb], mids[i % nb]))
    t_pool = timeit(lambda i: pool.submit(mids[i % nb], outs[i % nb]))
    def pair(i):
        op.submit(srcs[i % nb], mids[i % nb]); pool.submit(mids[i % nb], outs[i % nb])
    t_pair = timeit(pair)
    fop = hipref.make_conv(case, data, fuse_pool=2)
    t_fused = timeit(lambda i: fop.submit(srcs[i % nb], outs[i % nb]))
    print(json.dumps({"shape": [bs, hw, hw, ic, oc], "fused_ms": round(t_fused, 5), "fused_kernel": fop.info().kernel_name.decode(), "intermediate_MB": round(mids[0].numel() / 1e6, 1), "conv_ms": round(t_conv, 5), "pool_cold_ms": round(t_pool, 5),
                      "pair_ms": round(t_pair, 5), "pool_behind_conv_ms": round(t_pair - t_conv, 5)}))
