set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p $O
show() { python - "$1" <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'value', '%.4g' % l['value'], 'ms_per_step', round(l['ms_per_step'],4), 'kernel', round(l['roofline']['kernel_ms_avg'],4), 'frac', round(l['roofline']['frac'],4), l.get('preheat'))
for k, v in (l.get('configs') or {}).items():
    print('   ', k, v.get('kernel_ms_avg'), v.get('roofline', {}).get('frac'), v.get('ms_per_step'))
PY
}
time timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/full.json 2>$O/full.err; show $O/full.json
