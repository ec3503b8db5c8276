set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3g; mkdir -p $O
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 600 python tools/counters.py --config rtty > $O/ctr_rtty.log 2>&1; cat $O/ctr_rtty.log | grep -v amdgpu.ids
