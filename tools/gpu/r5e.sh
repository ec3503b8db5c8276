set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_dropin.py -x -q --timeout 120 --timeout-method=thread -p no:cacheprovider > $O/dropin.log 2>&1; echo "rc=$?" >> $O/dropin.log; grep -v amdgpu.ids $O/dropin.log | tail -6
timeout -s KILL 600 python bench.py --config 1200 --steps 10 > $O/bench_1200.json 2> $O/bench_1200.err; python - <<'PY'
import json
l=json.loads(open("gpurun_out/r5e/bench_1200.json").read().strip().splitlines()[-1])
print(json.dumps(l.get("legacy_dropin"), indent=1)); print(l["roofline"]["kernel_ms_avg"], l["roofline"]["frac"], l["oracle_mismatching_streams"])
PY
tail -3 $O/bench_1200.err
