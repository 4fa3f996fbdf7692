#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass over bench.py:

    export VCG_WGRAD_OVERLAP=0
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES \
        -d gpurun_out/pmc_m -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/pmc_mfma_util.py gpurun_out/pmc_m/m_counter_collection.csv 3 OUT.json > OUT.txt

util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is
summed over the 8 XCDs; MI355X_MICROARCH.md).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles: 32 per v_mfma_f32_32x32x16_bf16,
64 per v_mfma_f32_32x32x2_f32."""
import collections
import csv
import json
import re
import sys


def main():
    path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    per = collections.OrderedDict()
    seen = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            k = per.setdefault(name, {"mfma": 0.0, "gui": 0.0, "ns": 0, "n": 0})
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                k["mfma"] += float(r["Counter_Value"])
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                k["gui"] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen[r["Dispatch_Id"]] = 1
                k["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                k["n"] += 1
    rows = [(n, k) for n, k in per.items() if k["mfma"] > 0]
    rows.sort(key=lambda nk: -nk[1]["ns"])
    tot_m = sum(k["mfma"] for _, k in rows)
    tot_c = sum(k["gui"] / 8 * 1024 for _, k in rows)
    all_c = sum(k["gui"] / 8 * 1024 for k in per.values())
    print(f"MFMA-pipe utilisation per kernel ({steps} steps in the trace, one stream; kernels that issue MFMAs)")
    print(f"{'kernel':34s} {'launches/step':>13s} {'ms/step':>8s} {'MFMA pipe busy':>15s} {'clock GHz':>10s}")
    import os
    js = {"head": os.environ.get("VCG_HEAD", "unknown"), "kernels": {}}
    for n, k in rows:
        util = k["mfma"] / max(k["gui"] / 8 * 1024, 1)
        ghz = k["gui"] / 8 / max(k["ns"], 1)
        print(f"{n[:34]:34s} {k['n'] / steps:13.1f} {k['ns'] / steps / 1e6:8.3f} {100 * util:14.1f}% {ghz:10.2f}")
        js["kernels"][n] = {"ms_per_step": k["ns"] / steps / 1e6, "mfma_util": util}
    js["conv_kernels_mfma_util"] = tot_m / max(tot_c, 1)
    js["whole_step_mfma_util"] = tot_m / max(all_c, 1)
    print(f"\nall MFMA kernels together: {100 * js['conv_kernels_mfma_util']:.1f} % of their cycles; over every kernel of the step: "
          f"{100 * js['whole_step_mfma_util']:.1f} %")
    with open(out, "w") as fh:
        json.dump(js, fh, indent=1)


if __name__ == "__main__":
    main()
