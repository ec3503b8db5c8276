set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p $O
for L in profrows prof; do
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_$L.so timeout -s KILL 200 python tools/counters.py --config 1200noise --streams 256 > $O/ctr_$L.log 2>&1; echo "== $L"; grep -v amdgpu $O/ctr_$L.log | grep -i "sf_\|refine\|general\|restart\|bulk\|wall\|total" 
done
