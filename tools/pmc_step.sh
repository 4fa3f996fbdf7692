#!/bin/bash
# SQ wait / issue counters of every kernel of the training step (one stream): tools/pmc_step.sh <tag> -> gpurun_out/<tag>_pmc_step.txt
TAG=$1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
VCG_WGRAD_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcs_$TAG -o k -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-eager-baseline > $R/gpurun_out/pmcs_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/pmcs_$TAG.log; exit 3; }
python3 - <<PY
import csv, glob, collections
f=glob.glob("$R/gpurun_out/pmcs_$TAG/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:56]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[k]+=1
rows=sorted(agg.items(), key=lambda kv:-kv[1].get('GRBM_GUI_ACTIVE',0))
with open("$R/gpurun_out/${TAG}_pmc_step.txt","w") as o:
    o.write("# per kernel over 3 steps (1 stream): launches, GPU-active share, of the wave cycles: waiting (s_waitcnt/barrier) | issue-stalled | issuing; MFMA pipe busy share of GUI cycles x SIMDs; LDS conflict share\n")
    tot=sum(v.get('GRBM_GUI_ACTIVE',0) for _,v in rows)
    for k,v in rows[:60]:
        wc=max(v['SQ_WAVE_CYCLES'],1)
        line=f"{k:56s} n={cnt[k]:4d} gui={v.get('GRBM_GUI_ACTIVE',0)/tot*100:5.1f}% wait={v['SQ_WAIT_ANY']/wc*100:5.1f}% stall={v['SQ_WAIT_INST_ANY']/wc*100:5.1f}% issue={v['SQ_ACTIVE_INST_ANY']/wc*100:5.1f}% mfma={v['SQ_VALU_MFMA_BUSY_CYCLES']/max(v.get('GRBM_GUI_ACTIVE',1)/8*1024,1)*100:5.1f}% ldsconf={v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1)*100:5.1f}%"
        o.write(line+"\n")
print(open("$R/gpurun_out/${TAG}_pmc_step.txt").read())
PY
