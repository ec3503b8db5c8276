set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5m; mkdir -p $O
for c in ${CFGS:-1200 1200noise 12000 same rtty}; do
timeout -s KILL 240 python tools/gpu/overlap.py --config $c --steps ${STEPS:-40} 2>&1 | grep -v amdgpu | tee -a $O/overlap.log | tail -5
done
