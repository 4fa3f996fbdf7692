#!/usr/bin/env python3
"""Diagnostic: rounding error of the HIP convs (fwd / dgrad / wgrad) against float64, next to
PyTorch-CPU float32 on the same data.  GPU box: python tools/conv_accuracy.py"""
import importlib
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vae-cyclegan-implementation_amd")
dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.set_num_threads(16)


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


def run(cin, cout, h, w, n):
    mod = pkg.Networks.S(cin, cout)
    x = torch.randn(n, cin, h, w)
    g = torch.randn(n, cout, h, w)
    wt, b = mod.conv.weight.detach().clone(), mod.conv.bias.detach().clone()

    def cpu(dtype):
        xx = x.detach().clone().to(dtype).requires_grad_(True)
        ww = wt.detach().clone().to(dtype).requires_grad_(True)
        y = F.conv2d(F.pad(xx, (1, 1, 1, 1), mode="reflect"), ww, b.to(dtype))
        y.backward(g.to(dtype))
        return y.detach(), xx.grad, ww.grad
    y64, dx64, dw64 = cpu(torch.float64)
    y32, dx32, dw32 = cpu(torch.float32)
    mod = mod.to(dev)
    xg = x.detach().to(dev).requires_grad_(True)
    yg = mod(xg)
    yg.backward(g.to(dev))
    yh = pkg.ops.to_nchw_contiguous(yg.detach()).cpu()
    dxh = pkg.ops.to_nchw_contiguous(xg.grad).cpu() if pkg.ops.is_nhwc_view(xg.grad) else xg.grad.cpu()
    dwh = mod.conv.weight.grad.cpu()
    print(f"{cin:5d}->{cout:<5d} {h:3d}x{w:<3d} n={n}  K={9 * cin:6d} | fwd hip {rel(yh, y64):.2e} cpu32 {rel(y32, y64):.2e} | "
          f"dgrad hip {rel(dxh, dx64):.2e} cpu32 {rel(dx32, dx64):.2e} | wgrad hip {rel(dwh, dw64):.2e} cpu32 {rel(dw32, dw64):.2e}")


for cfg in [(64, 64, 32, 32, 2), (512, 512, 8, 8, 2), (1024, 1024, 16, 16, 2), (256, 128, 32, 32, 2), (2048, 1024, 4, 4, 2),
            # round 2: shapes that take the LDS-slab forward / data gradient and the row-ring weight gradient (maps >= 64 x 64)
            (32, 64, 128, 128, 2), (64, 128, 64, 96, 2), (128, 256, 64, 64, 1)]:
    run(*cfg)
