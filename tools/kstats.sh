#!/bin/bash
# kernel-stats of a short bench run (one stream): tools/kstats.sh <tag> [env assignments...]  -> gpurun_out/<tag>_kernel_stats.csv
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
env VCG_WGRAD_OVERLAP=0 "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o k -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-eager-baseline > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 3
cp $(find $R/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/${TAG}_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6/6
n=sum(int(r['Calls']) for r in rows)/6
print("$TAG: kernels %.2f ms/step, %.0f launches/step" % (tot, n))
for r in rows[:45]:
    print("%-70s %6.1f calls/step %8.3f ms/step %8.1f us" % (r['Name'][:70], int(r['Calls'])/6, float(r['TotalDurationNs'])/1e6/6, float(r['AverageNs'])/1e3))
PY
