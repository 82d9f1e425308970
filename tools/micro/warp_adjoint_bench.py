import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepvideocodec_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (h, sig) in ((128, 1.3), (256, 8.0)):
    src = torch.randn(32, 64, h, h, generator=g).to(dev)
    import torch.nn.functional as F
    flow = (F.interpolate(torch.randn(32, 2, h // 16, h // 16, generator=g) * sig, size=(h, h), mode="bilinear") + torch.randn(32, 2, h, h, generator=g) * sig * 0.5).to(dev)
    go = torch.randn(32, 64, h, h, generator=g).to(dev)
    for it in range(3):
        ops.flow_warp_backward(src, flow, go)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5):
        ops.flow_warp_backward(src, flow, go)
    e1.record(); torch.cuda.synchronize()
    z = torch.zeros_like(src); torch.cuda.synchronize()
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e2.record()
    for it in range(5):
        z = torch.zeros_like(src)
    e3.record(); torch.cuda.synchronize()
    print(f"{h}x{h} sigma {sig}: {e0.elapsed_time(e1)/5*1e3:8.1f} us per call (of which zero fill {e2.elapsed_time(e3)/5*1e3:6.1f})", flush=True)
