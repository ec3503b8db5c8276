set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p $O
show() { python - "$1" <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms_per_step', round(l['ms_per_step'],4), 'kernel', round(l['roofline']['kernel_ms_avg'],4), 'frac', round(l['roofline']['frac'],4))
PY
}
for rep in 1 2; do
timeout -s KILL 200 python bench.py --config 1200 --no-cpu --no-h2d --no-extra > $O/b_step$rep.json 2>$O/err; show $O/b_step$rep.json
MIFSK_BENCH_REGION_EVENTS=1 timeout -s KILL 200 python bench.py --config 1200 --no-cpu --no-h2d --no-extra > $O/b_region$rep.json 2>$O/err; show $O/b_region$rep.json
done
timeout -s KILL 300 python tools/gpu/abn.py --config 1200 --libs base,main --rounds 5 --steps 10 2>&1 | grep "ms/launch"
timeout -s KILL 200 python bench.py --config 1200 --steps 200 --no-cpu --no-h2d --no-extra > $O/b_step200.json 2>$O/err; show $O/b_step200.json
MIFSK_BENCH_REGION_EVENTS=1 timeout -s KILL 200 python bench.py --config 1200 --steps 200 --no-cpu --no-h2d --no-extra > $O/b_region200.json 2>$O/err; show $O/b_region200.json
