set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
export MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so
for c in ${CFGS:-12000 same rtty}; do
timeout -s KILL 300 python tools/counters.py --config $c > $O/ctr_$c.log 2>&1; grep -v amdgpu $O/ctr_$c.log | head -${HEAD:-42}
done
