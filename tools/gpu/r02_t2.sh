set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_t2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 --durations=8 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log
grep -E "^FAILED|^E  |passed|failed|rc=|s call|s setup" $O/gpu.log | head -40
