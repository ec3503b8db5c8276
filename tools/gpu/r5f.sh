set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
for c in ${CFGS:-12000 same rtty}; do
timeout -s KILL 600 python tools/gpu/abn.py --config $c --libs ${LIBS:-base,main} --rounds ${ROUNDS:-3} --steps ${STEPS:-5} > $O/ab_$c.log 2>&1; grep -v amdgpu.ids $O/ab_$c.log | grep "ms/launch\|DIFFER\|Error\|error" | tail -6
done
