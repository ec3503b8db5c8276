set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_t4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q --timeout 300 --durations=5 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log
grep -E "^FAILED|^E  |passed|failed|rc=|s call" $O/gpu.log | head -40
