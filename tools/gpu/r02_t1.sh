set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_t1; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q --timeout 120 > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log
grep -E "^FAILED|^E  |passed|failed|rc=" $O/parity.log | head -30
