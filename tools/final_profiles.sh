# Regenerates the committed profiles of a round on the GPU box:
# (PART=1: the bench line, kernel stats and PMC passes; PART=2: the other workloads, microbench, accuracy — two gpurun calls)
#   VCG_HEAD=$(git rev-parse --short HEAD) gpurun --timeout 1100 -- "VCG_HEAD=$VCG_HEAD R=r04 PART=1 bash tools/final_profiles.sh"
# (the box has no .git: the commit the passes were taken at travels in VCG_HEAD and ends up in every JSON's "head")
set -x
cd $GRAFT_REPO_ROOT
R=${R:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out
export VCG_HEAD=${VCG_HEAD:-unknown}
if [ "${PART:-all}" != "2" ]; then
python bench.py --steps 10 --warmup 3 > $O/${R}_bench.json 2> $O/${R}_bench.err || exit 1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ov -o ov -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-eager-baseline > $O/prof_ov.log 2>&1 || exit 2
export VCG_WGRAD_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_se -o se -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-eager-baseline > $O/prof_se.log 2>&1 || exit 3
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-roofline > $O/pmc_f.log 2>&1 || exit 4
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-roofline > $O/pmc_w.log 2>&1 || exit 5
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/pmc_m -o m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-baseline --no-roofline > $O/pmc_m.log 2>&1 || exit 6
unset VCG_WGRAD_OVERLAP
cd $GRAFT_REPO_ROOT
F=$(find $O/pmc_f -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_w -name '*counter_collection.csv' | head -1); M=$(find $O/pmc_m -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py $F $W 3 $O/${R}_pmc_step_traffic.json > $O/${R}_pmc_step_traffic.txt || exit 7
python tools/pmc_mfma_util.py $M 3 $O/${R}_pmc_mfma_util.json > $O/${R}_pmc_mfma_util.txt || exit 8
cp $(find $O/prof_ov -name '*kernel_stats.csv' | head -1) $O/${R}_kernel_stats.csv
cp $(find $O/prof_se -name '*kernel_stats.csv' | head -1) $O/${R}_kernel_stats_serial.csv
bash tools/pmc_step.sh $R > /dev/null 2>&1 || echo "pmc_step failed"
[ "${PART:-all}" = "1" ] && { echo PART 1 DONE; exit 0; }
fi
if [ "${PART:-all}" != "1" ]; then
python bench.py --workload vae --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline > $O/${R}_bench_vae.json 2> $O/${R}_bench_vae.err || exit 9
python bench.py --workload autoencoder --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline > $O/${R}_bench_ae.json 2> $O/${R}_bench_ae.err || exit 10
python tools/conv_bench.py --layers stem,d1,d2,d3,d4,r,mu,vdb,u1,u2,u3,u4,head,disc1 > $O/${R}_conv_microbench.txt 2>&1 || exit 11
python tools/conv_accuracy.py > $O/${R}_conv_accuracy.txt 2>&1 || exit 12
python tools/power_probe.py --seconds 3 > $O/${R}_power_probe.txt 2>&1 || echo "power probe failed"
VCG_BENCH_SHAPES=$O/${R}_step_shapes.txt python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-eager-baseline > /dev/null 2>> $O/${R}_bench.err || exit 13
: > $O/${R}_bench_other_archs.jsonl
for w in cycleaegan cycleae cyclevae doubleae doublevae aegan vaegan; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline >> $O/${R}_bench_other_archs.jsonl 2>> $O/${R}_bench.err || echo "FAILED $w"
done
fi
echo ALL DONE
