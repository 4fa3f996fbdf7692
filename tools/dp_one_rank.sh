# One-rank RCCL rehearsal of the data-parallel step on a one-GPU box: what the exchange machinery costs with nothing to exchange.
#   gpurun --timeout 900 -- 'bash tools/dp_one_rank.sh'
# columns: ms/step, exchange_exposed_ms, buckets launched from inside the backward, buckets launched after it
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('exchange_exposed_ms'), d.get('exchange',{}).get('buckets_launched_from_inside_backward'), d.get('exchange',{}).get('buckets_launched_after_backward'))"; }
if [ -n "$1" ]; then for v in "$@"; do run VCG_X=0 $v; run VCG_FORCE_DIST=1 $v; done; exit 0; fi
run VCG_X=0
run VCG_FORCE_DIST=1
run VCG_FORCE_DIST=1 VCG_DP_FROM_BACKWARD=0
run VCG_FORCE_DIST=1 VCG_BUCKET_MB=4096
run VCG_WGRAD_OVERLAP=0
run VCG_FORCE_DIST=1 VCG_WGRAD_OVERLAP=0
run VCG_FORCE_DIST=1 VCG_DP_NULL_EXCHANGE=1
