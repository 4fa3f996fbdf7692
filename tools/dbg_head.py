import importlib, os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vae-cyclegan-implementation_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, h = 2, 64
spec = ops.ConvSpec(64, 3, 7, 1, 3, True, 1)
for trial, (wstd, gkind) in enumerate(((0.117, "sign"), (0.117, "randn"), (0.05, "sign"), (0.117, "sign_sparse"))):
    w = torch.nn.Parameter(torch.randn(3, 64, 7, 7, device=dev) * wstd)
    b = torch.nn.Parameter(torch.zeros(3, device=dev))
    x = ops.to_nhwc(torch.rand(n, 64, h, h, device=dev)).requires_grad_(True)
    y = ops.conv_block(x, w, b, spec)
    s = 1.0 / (n * 3 * h * h)
    if gkind == "sign":
        g = torch.sign(torch.randn(n, 3, h, h, device=dev)) * s
    elif gkind == "sign_sparse":
        g = torch.sign(torch.randn(n, 3, h, h, device=dev)) * s * (torch.rand(n, 3, h, h, device=dev) > 0.3)
    else:
        g = torch.randn(n, 3, h, h, device=dev) * s
    y.backward(ops.to_nhwc(g))
    dx = x.grad
    xr = x.detach().contiguous().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (3, 3, 3, 3), mode="reflect"), w.detach(), b.detach())
    yr.backward(g)
    bad = ~torch.isfinite(dx)
    print(trial, wstd, gkind, "nonfinite dx", bad.sum().item(), "wmax", w.abs().max().item(), "gmax", g.abs().max().item())
    if bad.any():
        idx = bad.nonzero()
        pix = sorted({(int(a), int(c), int(d)) for a, _, c, d in idx.tolist()})
        print("  #pixels", len(pix), pix[:40])
    else:
        print("  dgrad err", ((dx - xr.grad).norm() / xr.grad.norm()).item())
