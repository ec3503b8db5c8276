set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
timeout -s KILL 400 python tools/gpu/abn.py --config 1200 --libs ${LIBS:-base,main} --rounds 5 --steps 10 > $O/ab_1200.log 2>&1; grep -v amdgpu.ids $O/ab_1200.log | tail -8
timeout -s KILL 300 python tools/gpu/abn.py --config 1200noise --libs ${LIBS:-base,main} --rounds 3 --steps 10 > $O/ab_1200noise.log 2>&1; grep -v amdgpu.ids $O/ab_1200noise.log | tail -8
if [ -n "${PARITY:-1}" ]; then
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py -x -q --timeout 60 --timeout-method=thread -k "${K:-workgroup}" -p no:cacheprovider > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; grep -v amdgpu.ids $O/parity.log | tail -4
fi
