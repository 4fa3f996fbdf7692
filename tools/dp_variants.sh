# Alternating same-box comparison of one-rank RCCL variants of the bench step (see tools/dp_one_rank.sh)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('exchange_exposed_ms'), d.get('exchange',{}).get('buckets_launched_from_inside_backward'), d.get('exchange',{}).get('buckets_launched_after_backward'))"; }
for i in 1 2; do
run VCG_X=0
run VCG_FORCE_DIST=1
run VCG_FORCE_DIST=1 VCG_DP_EMULATE_BUSY=8
run VCG_FORCE_DIST=1 VCG_DP_EMULATE_BUSY=8 VCG_DIR_STREAMS=0
run VCG_FORCE_DIST=1 VCG_DP_EMULATE_BUSY=24
run VCG_FORCE_DIST=1 VCG_DP_EMULATE_BUSY=24 VCG_DIR_STREAMS=0
done
