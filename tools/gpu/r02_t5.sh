set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_t5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -x -q --timeout 600 --durations=5 -k "multi_device or rccl or bench_line" > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log
grep -E "^FAILED|^E  |passed|failed|rc=|s call" $O/gpu.log | head -40
