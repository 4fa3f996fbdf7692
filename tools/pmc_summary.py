#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes over `bench.py` (one with FETCH_SIZE, one with WRITE_SIZE: the TCC block cannot hold
both) into per-kernel and per-family HBM/fabric traffic.

    rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/pmc_summary.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv STEPS OUT.json > OUT.txt

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced read stream, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.  The counters sit
on the L2's fabric side: Infinity-Cache hits are included, so this is an upper bound on HBM traffic.

Families (one C-ABI call = several device kernels) are recovered from the dispatch order with VCG_WGRAD_OVERLAP=0: a
Winograd input transform belongs to the weight gradient when k_wino_dy follows it, to the data gradient when the GEMM
after it is followed by k_wino_out_pad, else to the forward pass.
"""
import collections
import csv
import json
import re
import sys


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def load(path, counter):
    rows = collections.OrderedDict()
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            d = rows.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"]), "v": 0.0,
                                                        "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            d["v"] += float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


WGRAD = ("k_conv_wgrad", "k_wino_dy", "k_wino_wgrad_reduce", "k_wgrad_reduce", "k_wgrad_scatter", "k_slab_sum", "k_colsum",
         "k_wgrad_ring", "k_ring_sum", "k_ring_scatter")
DGRAD = ("k_conv_dgrad", "k_wino_out_pad", "k_wino_out_fold", "k_wino_fold", "k_fold_pad", "k_conv_thin<1>")
FWD = ("k_conv_fwd", "k_wino_out", "k_conv_thin<0>", "k_splitk_finish")


def families(seq):
    fam = [None] * len(seq)
    for i, d in enumerate(seq):
        n = d["name"]
        if n.startswith(WGRAD):
            fam[i] = "conv_wgrad"
        elif n.startswith(DGRAD):
            fam[i] = "conv_dgrad"
        elif n.startswith("k_wino_in"):
            nxt = [seq[j]["name"] for j in range(i + 1, min(i + 4, len(seq)))]
            fam[i] = "conv_wgrad" if nxt and nxt[0].startswith("k_wino_dy") else \
                     "conv_dgrad" if len(nxt) > 1 and nxt[1].startswith(("k_wino_out_pad", "k_wino_out_fold")) else "conv_fwd"
        elif n.startswith(("k_gemm_split", "k_gemm_planes")):
            nxt = seq[i + 1]["name"] if i + 1 < len(seq) else ""
            fam[i] = "conv_dgrad" if nxt.startswith(("k_wino_out_pad", "k_wino_out_fold")) else \
                     "conv_wgrad" if nxt.startswith("k_wino_wgrad_reduce") else "conv_fwd"
        elif n.startswith(("k_conv_slab", "k_kwfold")):        # data gradient when a fold of the padded image follows
            nxt = [seq[j]["name"] for j in range(i + 1, min(i + 3, len(seq)))]
            fam[i] = "conv_dgrad" if any(x.startswith("k_fold_pad") for x in nxt) else "conv_fwd"
        elif n.startswith(FWD):
            fam[i] = "conv_fwd"
        elif n.startswith("k_in_bwd") or n.startswith("k_in_partial<1>") or n.startswith("k_in_final<1>"):
            fam[i] = "in_bwd"
        elif n.startswith("k_in_"):
            fam[i] = "in_fwd"
        else:
            fam[i] = "other"
    return fam


def main():
    fpath, wpath, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rd, wr = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    per = collections.OrderedDict()
    for seq, key, scale in ((rd, "read", 2 * 1024.0), (wr, "write", 1024.0)):
        fam = families(seq)
        for d, f in zip(seq, fam):
            k = per.setdefault(d["name"], {"read": 0.0, "write": 0.0, "ns": 0, "n": 0, "fam": collections.Counter()})
            k[key] += d["v"] * scale
            if key == "read":
                k["ns"] += d["ns"]
                k["n"] += 1
                k["fam"][f] += 1
    fams = collections.OrderedDict()
    for seq, key, scale in ((rd, "read", 2 * 1024.0), (wr, "write", 1024.0)):
        for d, f in zip(seq, families(seq)):
            x = fams.setdefault(f, {"read": 0.0, "write": 0.0, "ns": 0})
            x[key] += d["v"] * scale
            if key == "read":
                x["ns"] += d["ns"]
    print(f"per step ({steps} steps in the trace, all kernels; read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB; fabric-side, an upper bound on HBM)")
    print(f"{'kernel':34s} {'launches':>8s} {'read MB':>10s} {'write MB':>10s} {'ms':>8s} {'GB/s':>8s}")
    tot = {"read": 0.0, "write": 0.0, "ns": 0}
    for name, k in sorted(per.items(), key=lambda kv: -(kv[1]["read"] + kv[1]["write"])):
        tot["read"] += k["read"]; tot["write"] += k["write"]; tot["ns"] += k["ns"]
        gbs = (k["read"] + k["write"]) / max(k["ns"], 1)
        print(f"{name[:34]:34s} {k['n'] / steps:8.1f} {k['read'] / steps / 1e6:10.1f} {k['write'] / steps / 1e6:10.1f} "
              f"{k['ns'] / steps / 1e6:8.3f} {gbs:8.0f}")
    print(f"{'all kernels':34s} {'':8s} {tot['read'] / steps / 1e6:10.1f} {tot['write'] / steps / 1e6:10.1f} {tot['ns'] / steps / 1e6:8.3f} "
          f"{(tot['read'] + tot['write']) / max(tot['ns'], 1):8.0f}")
    print("\nper family (C-ABI call groups)")
    import os
    js = {"steps": steps, "unit": "bytes per step", "head": os.environ.get("VCG_HEAD", "unknown"), "kernels": {
        name: {"read": k["read"] / steps, "write": k["write"] / steps, "launches": k["n"] / steps, "ms": k["ns"] / steps / 1e6}
        for name, k in per.items()}, "note": "read = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024; "
          "fabric-side counters (Infinity-Cache hits included); profiled with VCG_WGRAD_OVERLAP=0", "families": {}}
    for f, x in fams.items():
        print(f"{f:12s} read {x['read'] / steps / 1e6:9.1f} MB  write {x['write'] / steps / 1e6:9.1f} MB  kernel time {x['ns'] / steps / 1e6:7.3f} ms  "
              f"{(x['read'] + x['write']) / max(x['ns'], 1):6.0f} GB/s")
        js["families"][f] = {"read": x["read"] / steps, "write": x["write"] / steps, "kernel_ms": x["ns"] / steps / 1e6}
    js["all"] = {"read": tot["read"] / steps, "write": tot["write"] / steps, "kernel_ms": tot["ns"] / steps / 1e6}
    with open(out, "w") as fh:
        json.dump(js, fh, indent=1)


if __name__ == "__main__":
    main()
