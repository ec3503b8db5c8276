set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -s KILL 1200 python -m pytest tests -q -m gpu --timeout 300 -p no:cacheprovider > $O/gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh ${TAG:-r06} "${CONFIGS:-1200 1200noise rtty 12000 same}" > $O/profile.log 2>&1
timeout -s KILL 600 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l = json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print('1200', l['ms_per_step'], l['roofline']['kernel_ms_avg'], l['roofline']['frac'], l['payload_roundtrip_ok_streams'], l['roofline'].get('traffic'))
for k, v in l['configs'].items():
    print(k, v['kernel_ms_avg'], v['roofline']['frac'], v['payload_roundtrip_ok_streams'], v['launch'].get('chain_groups'), v['launch'].get('chain_chunks'), v['roofline'].get('traffic'))
PY
