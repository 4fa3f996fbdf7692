#!/bin/bash
# A/B of one environment switch on one box: tools/ab_bench.sh VAR "0 1 0 1" [bench args...] -> lines "VAR=v images/s ms/step"
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-eager-baseline "$@" 2>>gpurun_out/ab_bench.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'])" || exit 1
done
