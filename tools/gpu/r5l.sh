set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p $O
show() { python - "$1" <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'value', '%.4g' % l['value'], 'ms_per_step', round(l['ms_per_step'],4), 'kernel', round(l['roofline']['kernel_ms_avg'],4), 'frac', round(l['roofline']['frac'],4), l.get('pipeline',{}).get('hbm_frac_of_the_timed_passes'), l['payload_roundtrip_ok_streams'], l.get('oracle_mismatching_streams'))
for k, v in (l.get('configs') or {}).items():
    print('   ', k, v.get('kernel_ms_avg'), v.get('roofline', {}).get('frac'), v.get('ms_per_step'), v.get('payload_roundtrip_ok_streams'), v.get('oracle_mismatching_streams'), v.get('error'))
PY
}
timeout -s KILL 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-h2d --no-extra --pipeline 1 > $O/p1.json 2>$O/err || tail -5 $O/err; show $O/p1.json
timeout -s KILL 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-h2d --no-extra > $O/p3.json 2>$O/err || tail -5 $O/err; show $O/p3.json
time timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/full.json 2>$O/full.err; show $O/full.json
timeout 300 python -m pytest tests/test_gpu_dist.py -x -q -p no:cacheprovider 2>&1 | tail -2
