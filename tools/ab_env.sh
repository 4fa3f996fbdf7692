#!/bin/bash
# same-box A/B of two environments, alternating: tools/ab_env.sh "<env A>" "<env B>" [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-eager-baseline 2>>gpurun_out/ab_bench.err \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], d['ms_per_step'])" || exit 1
  done
done
