# round 4: whole-batch oracle parity in the full-size tests and in the bench line
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b; mkdir -p $O
nproc; free -g | head -2
( time timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_files.py -x -q -s --timeout 900 ) > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log; grep -v amdgpu.ids $O/gpu.log | tail -25
( time timeout 900 python bench.py ) > $O/bench.json 2>$O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r4b/bench.json').read().strip().splitlines()[0])
print('1200', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], 'oracle', l.get('oracle_mismatching_streams'), l.get('oracle',{}).get('seconds'))
for k,v in l.get('configs',{}).items():
    print(k, v.get('kernel_ms_avg'), v.get('roofline',{}).get('frac'), 'oracle', v.get('oracle_mismatching_streams'), (v.get('oracle') or {}).get('seconds'), v.get('error'))
print(json.dumps(l['configs']['same'].get('payload_by_condition'))[:1500])
PY
