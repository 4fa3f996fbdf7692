#!/usr/bin/env python3
"""Diagnostic: where does a conv launch spend its time?

Builds a PRIVATE copy of the library with -DVCG_STAMP (tools/_build/libvcg_stamp.so; libvcg.so never carries the
stamps), runs one conv kernel of a layer of the 256x256 step and reads the per-workgroup stamps:
  launch skew, workgroup duration spread, prologue / main loop / epilogue split, the shader clock the chip holds
  (delta s_memtime / delta s_memrealtime x 100 MHz), and the workgroups-per-CU placement.
    python tools/stamp_probe.py [--layers d2,r] [--kinds fwd,dgrad,wgrad] [--warm 30]
"""
import argparse
import collections
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("vae-cyclegan-implementation_amd")
native, ops = pkg._native, pkg.ops
from conv_bench import LAYERS  # noqa: E402


def build_stamped():
    out_dir = os.path.join(ROOT, "tools", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libvcg_stamp.so")
    srcs = [os.path.join(native.CSRC, s) for s in native.SOURCES]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call([native.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-DVCG_STAMP", "-o", out] + srcs)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="d2,r")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--warm", type=int, default=30, help="back-to-back launches before the stamped one")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--dbg", default="", help="time instead of stamping: comma list of 'label[:ENV=VALUE[:ENV=VALUE]]' arms, "
                    "each run with those environment switches of the diagnostic build (VCG_NO_WINOGRAD=1, VCG_WGRAD_BM=256)")
    args = ap.parse_args()
    path = build_stamped()
    if args.build_only:
        return
    native.LIB_PATH = path                       # the package now binds the stamped copy
    lib = native.lib()
    lib.vcg_debug_set_stamp.restype, lib.vcg_debug_set_stamp.argtypes = ctypes.c_int, [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    stamps = torch.zeros(8 * (1 << 16), dtype=torch.int64, device=dev)

    for name in args.layers.split(","):
        cin, cout, k, s, pad, ups, h, cphys = LAYERS[name]
        spec = ops.ConvSpec(cin, cout, k, s, pad, True, ups)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        bias = torch.zeros(cout, device=dev)
        x = ops.as_phys(ops.to_nhwc(torch.randn(args.batch, cphys, h, h, device=dev)))
        cd = spec.desc(args.batch, h, h)
        ho, wo = spec.out_hw(h, h)
        y = torch.empty(args.batch * ho * wo * spec.cout_pitch, device=dev)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.zeros_like(w)
        wf = spec.packed(w)
        nws = max(lib.vcg_conv_fwd_workspace(cd), lib.vcg_conv_dgrad_workspace(cd), lib.vcg_conv_wgrad_workspace(cd), 16)
        ws = torch.empty(nws // 4 + 16, device=dev)
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        calls = {
            "fwd": lambda: lib.vcg_conv_fwd(P(x), P(wf), P(bias), P(y), cd, P(ws), nws, stream),
            "dgrad": lambda: lib.vcg_conv_dgrad(P(dy), P(wf), P(dx), cd, P(ws), nws, stream),
            "wgrad": lambda: lib.vcg_conv_wgrad(P(x), P(dy), P(dw), None, cd, P(ws), nws, stream),
        }
        flops = 2.0 * args.batch * ho * wo * spec.cout_pitch * k * k * ups * ups * spec.cin_pitch
        for kind in args.kinds.split(","):
            fn = calls[kind]
            lib.vcg_debug_set_stamp(None)
            if args.dbg:
                for item in args.dbg.split(","):
                    _, _, envs = item.partition(":")              # "label[:NAME=VALUE[:NAME=VALUE]]"
                    for kv in filter(None, envs.split(":")):
                        os.environ[kv.split("=")[0]] = kv.split("=")[1]
                    for _ in range(5):
                        native.check(fn(), kind)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        native.check(fn(), kind)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 100.0
                    print(f"{name} {kind} dbg={item:24s}: {us:8.1f} us  {flops / us * 1e-6:6.1f} TF", flush=True)
                continue
            for _ in range(args.warm):
                native.check(fn(), kind)
            torch.cuda.synchronize()
            stamps.zero_()
            torch.cuda.synchronize()
            lib.vcg_debug_set_stamp(ctypes.c_void_p(stamps.data_ptr()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.check(fn(), kind)
            e1.record()
            torch.cuda.synchronize()
            lib.vcg_debug_set_stamp(None)
            st = stamps.cpu().numpy().reshape(-1, 8)
            st = st[st[:, 0] != 0]
            report(f"{name} {kind}", st, e0.elapsed_time(e1) * 1e3)


def report(tag, st, event_us):
    t0, t1, t2, t3 = (st[:, i].astype(np.float64) * 0.01 for i in range(4))       # 100 MHz ticks -> us
    base = t0.min()
    dur = t3 - t0
    clk = (st[:, 5] - st[:, 4]).astype(np.float64) / np.maximum(st[:, 3] - st[:, 0], 1) * 100e6 / 1e9
    hw = st[:, 6]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    xcc = st[:, 7] & 0xF
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = collections.Counter(cuid.tolist())
    wave_slot, simd = hw & 0xF, (hw >> 4) & 0x3
    slots_by_cu = collections.defaultdict(list)
    for c_, w_, s_, a_, b_ in zip(cuid.tolist(), wave_slot.tolist(), simd.tolist(), t0.tolist(), t3.tolist()):
        slots_by_cu[c_].append((round(a_ - base, 1), round(b_ - base, 1), s_, w_))
    print("   wave-0 (start, end, simd, slot) of the workgroups on 3 CUs: " +
          " | ".join(str(sorted(v)[:6]) for v in list(slots_by_cu.values())[:3]))
    print(f"   slot histogram of wave 0: {dict(collections.Counter(wave_slot.tolist()))}")
    print(f"== {tag}: {len(st)} workgroups, event {event_us:.1f} us, span {t3.max() - base:.1f} us, "
          f"distinct CUs {len(per_cu)}, WG/CU min/max {min(per_cu.values())}/{max(per_cu.values())}")
    q = lambda a: " ".join(f"{v:8.1f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))  # noqa: E731
    print(f"   start offset   [min p10 p50 p90 max] us: {q(t0 - base)}")
    print(f"   end offset                            : {q(t3 - base)}")
    print(f"   WG duration                           : {q(dur)}")
    print(f"   prologue                              : {q(t1 - t0)}")
    print(f"   main loop                             : {q(t2 - t1)}")
    print(f"   epilogue                              : {q(t3 - t2)}")
    print(f"   shader clock GHz                      : {q(clk)}")
    # how many workgroups are alive over time (coarse)
    grid = np.linspace(base, t3.max(), 11)
    alive = [int(((t0 <= g) & (t3 > g)).sum()) for g in grid]
    print(f"   alive at 0..100% of span              : {alive}")
    by_xcc = collections.defaultdict(list)
    for x_, d_ in zip(xcc.tolist(), dur.tolist()):
        by_xcc[x_].append(d_)
    print("   mean WG duration per XCD              : " + " ".join(f"{k}:{np.mean(v):.0f}" for k, v in sorted(by_xcc.items())))


if __name__ == "__main__":
    main()
