import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from deepvideocodec_amd import ops
torch.manual_seed(0)
for (cin, cout, H, W, use_res) in ((64, 64, 544, 960, False), (64, 64, 544, 960, True), (128, 64, 544, 960, False)):
    x = torch.randn(1, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(cin * 9)
    b = torch.randn(cout, device="cuda")
    res = torch.randn(1, cout, H, W, device="cuda") if use_res else None
    want = F.conv2d(x, w, b, padding=1) + (res if use_res else 0)
    for it in range(2):
        got = ops.conv2d(x, w, b, precision=3, residual=res)
        torch.cuda.synchronize()
        d = (got - want).abs()
        bad = ~(d < 1e-3)
        print(f"cin {cin} res {use_res} run {it}: bad {int(bad.sum())} of {bad.numel()}, nan {int(torch.isnan(got).sum())}, max err (finite) {float(d[~torch.isnan(d)].max()):.3e}")
        if bad.any():
            idx = bad.nonzero()
            print("  first bad:", idx[:5].tolist(), " last bad:", idx[-3:].tolist())
            print("  bad per channel (first 8):", bad[0].sum((1, 2))[:8].tolist())
            rows = bad[0].any(0).any(1).nonzero().flatten()
            cols = bad[0].any(0).any(0).nonzero().flatten()
            print("  bad rows:", rows[:20].tolist(), "... n =", len(rows), " bad cols n =", len(cols), cols[:10].tolist())
