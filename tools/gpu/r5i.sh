set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
for c in ${CFGS:-12000 same}; do
timeout -s KILL 600 python tools/gpu/abn.py --config $c --libs ${LIBS:-base,main} --rounds ${ROUNDS:-3} --steps ${STEPS:-5} > $O/ab_$c.log 2>&1; grep -v amdgpu.ids $O/ab_$c.log | grep "ms/launch\|DIFFER\|results\|Error\|error" | tail -6
done
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q --timeout 300 --timeout-method=thread -p no:cacheprovider ${KARG:-} > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log; grep -v amdgpu.ids $O/parity.log | tail -6
