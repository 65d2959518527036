"""prints the device pointers a direct-kernel op would use (no launch; debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["DFX_DEBUG_PTRS"] = "1"
import torch
assert torch.cuda.is_available()
import cases as C
import hipref
for case in (C.ConvCase("d64", 2, 64, 12, 10, 64, 64, dst_dt=C.S8, relu1=False),
             C.ConvCase("res3", 2, 128, 14, 14, 128, 512, dst_dt=C.S32)):
    data = C.generate(case)
    op = hipref.make_conv(case, data, force_variant=3)
    src = torch.from_numpy(data["src"]).cuda()
    dst = torch.empty(op.dst_shape, dtype=torch.int8 if case.dst_dt == C.S8 else torch.int32, device="cuda")
    dst.view(torch.uint8).fill_(0xCD)
    torch.cuda.synchronize()
    print(case.name, "src %#x..%#x dst %#x..%#x" % (src.data_ptr(), src.data_ptr() + src.numel(), dst.data_ptr(),
                                                    dst.data_ptr() + dst.numel() * dst.element_size()), flush=True)
    op.close()
