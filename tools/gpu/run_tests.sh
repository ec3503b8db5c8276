# find the hanging case: sequential, verbose, hard per-test and overall limits
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4e; mkdir -p $O
timeout -s KILL ${LIMIT:-300} python -m pytest ${TESTS:-tests/test_gpu_slabs.py} -x -v --timeout 40 --timeout-method=thread -k "${K:-workgroup}" -p no:cacheprovider > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log; grep -v amdgpu.ids $O/gpu.log | grep -v "PASSED\|SKIPPED" | tail -40; grep -c PASSED $O/gpu.log; grep -E "PASSED|SKIPPED" $O/gpu.log | tail -3
