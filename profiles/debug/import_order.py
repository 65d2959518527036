import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import cases as C, hipref
case = C.CONFIG2
data = C.generate(case)
op = hipref.make_conv(case, data)          # HIP used by libdfx_hip.so before any torch.cuda call
import torch
x = torch.from_numpy(data["src"]).cuda()
print("ok", op.info().kernel_name, x.shape)
