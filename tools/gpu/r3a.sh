# round-3 check A: whole GPU suite, the driver's bench line (all four configs), the long-window
# engine comparison
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log; tail -5 $O/gpu.log
timeout 400 python bench.py --no-h2d > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r3a/bench.json').read().strip().splitlines()[-1])
print('1200', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'], l.get('cpu_baseline',{}).get('mismatching_streams'))
for k,v in l.get('configs',{}).items():
    print(k, v.get('kernel_ms_avg'), v.get('roofline',{}).get('frac'), v.get('payload_roundtrip_ok_streams'), v.get('error'))
PY
timeout 200 python tools/gpu/eng50.py > $O/eng50.log 2>&1; tail -4 $O/eng50.log
