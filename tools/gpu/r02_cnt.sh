set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export MIFSK_LIBRARY=$GRAFT_REPO_ROOT/minimodem_amd/libmifsk_prof.so
for C in rtty 12000 same; do timeout 120 python tools/counters.py --config $C 2>&1 | grep -v amdgpu.ids; done
