set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c; mkdir -p $O
timeout 500 python bench.py --no-cpu --no-h2d > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r3c/bench.json').read().strip().splitlines()[-1])
print('1200', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'])
for k,v in l.get('configs',{}).items():
    print(k, v.get('kernel_ms_avg'), v.get('roofline',{}).get('frac'), v.get('payload_roundtrip_ok_streams'), v.get('error'))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -q --timeout 600 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log; tail -4 $O/gpu.log | cut -c1-300
