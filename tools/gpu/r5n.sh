set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5final; mkdir -p $O
time (timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err); echo "bench rc=$?"
python - <<'PY'
import json
l = json.loads(open('gpurun_out/r5final/bench.json').read().strip().splitlines()[-1])
print('1200 value %.4g' % l['value'], l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'], l.get('oracle_mismatching_streams'))
for k, v in l['configs'].items():
    print(k, 'value %.4g' % v['value'], v.get('ms_per_step'), v.get('kernel_ms_avg'), v.get('roofline', {}).get('frac'), v.get('payload_roundtrip_ok_streams'), v.get('oracle_mismatching_streams'), v.get('error'))
PY
