set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
export MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so
timeout -s KILL 200 python tools/counters.py --config ${CFG:-1200noise} --streams ${NS:-256} > $O/ctr.log 2>&1; grep -v amdgpu $O/ctr.log | head -${HEAD:-40}
