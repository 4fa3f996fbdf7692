cd $GRAFT_REPO_ROOT
for w in cycleaegan cycleae cyclevae doubleae doublevae aegan vaegan; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline >> gpurun_out/bench_others.jsonl 2>> gpurun_out/bench_others.err || echo "FAILED $w"
done
python bench.py --steps 10 --warmup 3 > gpurun_out/bench32.json 2> gpurun_out/bench32.err
echo done
