set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_t6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_txdev.py -x -q --timeout 300 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log
grep -E "^FAILED|^E  |passed|failed|rc=" $O/gpu.log | head
