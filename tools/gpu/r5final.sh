# end-of-round evidence: profiles of the five bench workloads, the default bench line, copy interference
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5final; mkdir -p $O
bash tools/profile_round.sh ${TAG:-r05} "${CFGS:-1200 1200noise 12000 same rtty}" > $O/profile.log 2>&1; tail -3 $O/profile.log
timeout -s KILL 300 python tools/gpu/copy_interference.py > $O/copy_interference.log 2>&1; grep -v amdgpu $O/copy_interference.log
timeout -s KILL 900 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l = json.loads(open('gpurun_out/r5final/bench.json').read().strip().splitlines()[-1])
print('1200', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'], l.get('oracle_mismatching_streams'))
for k, v in l['configs'].items():
    print(k, v.get('kernel_ms_avg'), v.get('roofline', {}).get('frac'), v.get('payload_roundtrip_ok_streams'), v.get('oracle_mismatching_streams'), v.get('error'))
PY
