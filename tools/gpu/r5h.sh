set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method=thread -p no:cacheprovider ${KARG:-} > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; grep -v amdgpu.ids $O/parity.log | tail -${TAIL:-8}
