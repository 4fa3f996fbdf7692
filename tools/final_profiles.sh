set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
python bench.py --steps 10 --warmup 3 > $O/bench28.json 2> $O/bench28.err || exit 1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof18 -o p18 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof18.log 2>&1 || exit 2
export VCG_WGRAD_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof19 -o p19 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof19.log 2>&1 || exit 3
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_f.log 2>&1 || exit 4
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_w.log 2>&1 || exit 5
unset VCG_WGRAD_OVERLAP
cd $GRAFT_REPO_ROOT
python bench.py --workload vae --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_vae2.json 2> $O/bench_vae2.err || exit 6
python bench.py --workload autoencoder --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ae2.json 2> $O/bench_ae2.err || exit 7
python tools/conv_bench.py --layers stem,d1,d2,d3,d4,r,mu,vdb,u1,u2,u3,u4,head,disc1 > $O/microbench2.txt 2>&1 || exit 8
echo ALL DONE
