cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('exchange_exposed_ms'), d.get('exchange',{}).get('buckets_launched_from_inside_backward'), d.get('exchange',{}).get('buckets_launched_after_backward'))"; }
for i in 1 2; do
run VCG_X=0
run VCG_FORCE_DIST=1 VCG_BUCKET_MB=64
run VCG_FORCE_DIST=1 VCG_BUCKET_MB=128
run VCG_FORCE_DIST=1 VCG_BUCKET_MB=256
done
