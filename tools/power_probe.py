#!/usr/bin/env python3
"""Socket power and shader clock while the step's GEMM shapes run back to back (the "power-bound" reading of the
in-kernel stamps rested on s_memtime / s_memrealtime alone).

A side thread samples the amdgpu hwmon / sysfs files of the card (power1_average or power1_input in microwatts, freq1_input in
Hz, pp_dpm_sclk's starred level) every ~10 ms while the main thread keeps one layer's forward (or forward + backward) in
flight for a few seconds; the same is done for an idle card, for an HBM-bound kernel (the fused Adam over 140 M parameters) and
for each of --layers.  Prints one line per phase: samples, power mean / max (W), sclk mean / min / max (MHz), per-launch time.

    python tools/power_probe.py [--seconds 3] [--layers r,d2,d3,d4] [--fwd-only]
Falls back to `rocm-smi` / `amd-smi` one-shot readings (slow: one sample per ~0.3 s) when no sysfs file is readable, and says so.
"""
import argparse
import glob
import importlib
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def find_sources():
    src = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        dev = os.path.join(card, "device")
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "power1_cap"):
                p = os.path.join(hw, name)
                if os.path.exists(p) and os.access(p, os.R_OK):
                    src.setdefault(card, {})[name] = p
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent", "mem_busy_percent"):
            p = os.path.join(dev, name)
            if os.path.exists(p) and os.access(p, os.R_OK):
                src.setdefault(card, {})[name] = p
    return src


def read_file(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None


def starred_mhz(text):
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            tok = line.split(":")[1].strip().split("Mhz")[0].split("MHz")[0]
            try:
                return float(tok)
            except ValueError:
                return None
    return None


class Sampler(threading.Thread):
    def __init__(self, files, period=0.01):
        super().__init__(daemon=True)
        self.files, self.period = files, period
        self.rows, self.stop_flag = [], False

    def run(self):
        while not self.stop_flag:
            row = {"t": time.time()}
            for k in ("power1_average", "power1_input"):
                if k in self.files:
                    v = read_file(self.files[k])
                    if v and v.isdigit():
                        row["W"] = int(v) * 1e-6
                        break
            if "freq1_input" in self.files:
                v = read_file(self.files["freq1_input"])
                if v and v.isdigit():
                    row["sclk"] = int(v) * 1e-6
            if "sclk" not in row and "pp_dpm_sclk" in self.files:
                m = starred_mhz(read_file(self.files["pp_dpm_sclk"]))
                if m:
                    row["sclk"] = m
            if "freq2_input" in self.files:
                v = read_file(self.files["freq2_input"])
                if v and v.isdigit():
                    row["mclk"] = int(v) * 1e-6
            if "gpu_busy_percent" in self.files:
                v = read_file(self.files["gpu_busy_percent"])
                if v and v.isdigit():
                    row["busy"] = int(v)
            self.rows.append(row)
            time.sleep(self.period)


def smi_once():
    out = {}
    for cmd in (["rocm-smi", "--showpower", "--showclocks", "--json"], ["amd-smi", "metric", "-p", "-c", "--json"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            out[" ".join(cmd)] = (r.returncode, (r.stdout or r.stderr)[:1500])
        except Exception as e:  # noqa
            out[" ".join(cmd)] = (-1, repr(e))
    return out


def summarize(name, rows, per_launch_us=None, extra=""):
    w = [r["W"] for r in rows if "W" in r]
    s = [r["sclk"] for r in rows if "sclk" in r]
    m = [r["mclk"] for r in rows if "mclk" in r]
    b = [r["busy"] for r in rows if "busy" in r]

    def st(v, f="{:7.1f}"):
        return "   n/a " if not v else f.format(sum(v) / len(v))
    line = (f"{name:28s} samples {len(rows):5d} | power mean {st(w)} max {max(w) if w else float('nan'):7.1f} W | sclk mean {st(s, '{:7.0f}')} "
            f"min {min(s) if s else float('nan'):7.0f} max {max(s) if s else float('nan'):7.0f} MHz | mclk {st(m, '{:6.0f}')} | busy {st(b, '{:4.0f}')} %")
    if per_launch_us is not None:
        line += f" | {per_launch_us:8.1f} us/iter"
    print(line + extra, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--layers", default="r,d2,d3,d4,d1,u1")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--fwd-only", action="store_true")
    args = ap.parse_args()
    src = find_sources()
    print("sysfs sources:", json.dumps({c: sorted(v) for c, v in src.items()})[:400], "...", flush=True)
    # the box is a slice of an 8-GPU host: every card's hwmon is readable, one card is ours.  Match HIP's PCI address
    # (domain:bus:device.function) against the sysfs device links.
    files, mine = {}, None
    try:
        pr = torch.cuda.get_device_properties(0)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        for card in src:
            real = os.path.realpath(os.path.join(card, "device"))
            if want in real:
                mine = card
        print("HIP device 0 is PCI", want, "->", mine, flush=True)
    except Exception as e:  # noqa
        print("could not read the PCI address of HIP device 0:", repr(e))
    if mine is None and src:
        # fall back: the card whose busy counter moves while a kernel loop runs
        x = torch.empty(1 << 28, device="cuda:0")
        before = {c: read_file(f.get("gpu_busy_percent", "")) for c, f in src.items()}
        t0 = time.time()
        while time.time() - t0 < 1.5:
            x.add_(1.0)
        torch.cuda.synchronize()
        after = {c: read_file(f.get("gpu_busy_percent", "")) for c, f in src.items()}
        print("busy% before / after a 1.5 s kernel loop:", before, after, flush=True)
        cand = [c for c in src if (after[c] or "0").isdigit() and int(after[c] or 0) > int(before[c] or 0) + 20]
        mine = cand[0] if len(cand) == 1 else None
        del x
    files = src.get(mine, {}) if mine else {}
    if "power1_cap" in files:
        print("power cap:", read_file(files["power1_cap"]), "uW")
    if not files:
        print("no readable sysfs telemetry; one-shot SMI readings instead:", json.dumps(smi_once(), indent=1)[:4000])
    pkg = importlib.import_module("vae-cyclegan-implementation_amd")
    import conv_bench
    ops = pkg.ops
    dev = torch.device("cuda:0")

    def phase(name, body, sync_every=8):
        body()
        torch.cuda.synchronize()
        smp = Sampler(files)
        smp.start()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < args.seconds:
            for _ in range(sync_every):
                body()
                n += 1
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        smp.stop_flag = True
        smp.join()
        rows = smp.rows[len(smp.rows) // 5:]          # drop the ramp
        summarize(name, rows, e0.elapsed_time(e1) * 1e3 / n)

    smp = Sampler(files)
    smp.start()
    time.sleep(1.0)
    smp.stop_flag = True
    smp.join()
    summarize("idle", smp.rows)

    big = torch.empty(140_000_000, device=dev).normal_()
    g, m, v = torch.randn_like(big), torch.zeros_like(big), torch.zeros_like(big)
    phase("adam 140M (HBM-bound)", lambda: ops.adam_step_flat(big, g, m, v, 1, 2e-4, 0.5, 0.999, 1e-8))
    del big, g, m, v
    for name in args.layers.split(","):
        cin, cout, k, s, pad, ups, h, cphys = conv_bench.LAYERS[name]
        spec = ops.ConvSpec(cin, cout, k, s, pad, True, ups, epi_act=ops.ACT_RELU, norm=False)
        w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device=dev) * 0.05)
        b = torch.nn.Parameter(torch.zeros(cout, device=dev))
        x = ops.to_nhwc(torch.randn(args.batch, cphys, h, h, device=dev)).requires_grad_(True)

        def fwd():
            with torch.no_grad():
                ops.conv_block(x, w, b, spec)
        phase(f"{name} forward", fwd)
        if not args.fwd_only:
            y = ops.conv_block(x, w, b, spec)
            gy = ops.to_nhwc(torch.randn(tuple(y.shape), device=dev))

            def fb():
                x.grad = None
                ops.conv_block(x, w, b, spec).backward(gy)
            phase(f"{name} forward + backward", fb, sync_every=4)
    if os.environ.get("VCG_POWER_STEP", "1") != "0":
        # the whole training step, as bench.py runs it
        model = pkg.Networks.CycleVAEGAN(latent_dim=64, paired=False).to(dev).train()
        model.configure_optimizers(lr=2e-4)
        model.configure_loss()
        xb = ops.rand_uniform((args.batch, 3, 256, 256), dev, seed=1, offset=0)
        yb = ops.rand_uniform((args.batch, 3, 256, 256), dev, seed=1, offset=1 << 22)
        phase("cyclevaegan training_step", lambda: model.training_step({"x": xb, "y": yb}), sync_every=2)


if __name__ == "__main__":
    main()
