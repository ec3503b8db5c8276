set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3n; mkdir -p $O
for c in 12000 same; do
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 600 python tools/counters.py --config $c > $O/ctr_$c.log 2>&1; grep -v "amdgpu.ids\|XCD " $O/ctr_$c.log
done
