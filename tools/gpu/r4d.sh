# round 4: the workgroup engine's resumable twin -- slabs and chained launches on both engines
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4d; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_chain.py -x -q --timeout 900 -n 4 ) > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log; grep -v amdgpu.ids $O/gpu.log | tail -40
