set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 600 python tools/counters.py --config same 2>&1 | grep -v amdgpu
