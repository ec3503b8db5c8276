# round-3 check B: whole GPU suite (no -x: see every failure), the driver's bench line
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log; tail -25 $O/gpu.log | cut -c1-300
timeout 500 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r3b/bench.json').read().strip().splitlines()[-1])
print('1200', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'], l.get('cpu_baseline',{}).get('mismatching_streams'))
for k,v in l.get('configs',{}).items():
    print(k, v.get('kernel_ms_avg'), v.get('roofline',{}).get('frac'), v.get('payload_roundtrip_ok_streams'), v.get('error'))
print(json.dumps(l.get('h2d_inclusive'), indent=1))
PY
