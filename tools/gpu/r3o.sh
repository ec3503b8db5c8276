set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3o; mkdir -p $O
timeout 500 python bench.py --no-cpu --no-h2d > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r3o/bench.json').read().strip().splitlines()[-1])
print('1200', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'])
for k,v in l.get('configs',{}).items():
    print(k, v.get('kernel_ms_avg'), v.get('roofline',{}).get('frac'), v.get('payload_roundtrip_ok_streams'), v.get('launch',{}).get('lds_bytes_per_workgroup'), v.get('launch',{}).get('workgroups_per_cu'), v.get('error'))
PY
for n in 3000 4096 5000 6144; do timeout 200 python bench.py --no-cpu --config rtty --streams $n --steps 3 > $O/rtty_$n.json 2>>$O/bench.err; python -c "
import json; l=json.loads(open('$O/rtty_$n.json').read().strip().splitlines()[-1]); print('rtty n=$n', l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['roofline']['launch']['workgroups_per_cu'])"; done
