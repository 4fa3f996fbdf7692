#!/bin/bash
# SQ counters of the kernels of a conv_bench run: tools/pmc_kernel.sh <tag> "<conv_bench args>" -> gpurun_out/<tag>_pmc.txt
TAG=$1; ARGS=$2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_$TAG -o k -- python3 $R/tools/conv_bench.py $ARGS > $R/gpurun_out/pmc_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$TAG.log; exit 3; }
python3 - <<PY
import csv, glob, collections
f=glob.glob("$R/gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[k]+=1
with open("$R/gpurun_out/${TAG}_pmc.txt","w") as o:
    for k,v in agg.items():
        n=max(cnt[k],1)
        line=f"{k:60s} n={n:3d} "+" ".join(f"{c.replace('SQ_','')}={v[c]/n:.3g}" for c in sorted(v))
        print(line); o.write(line+"\n")
PY
