set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ctr; mkdir -p $O
for c in ${CFGS:-rtty same}; do
  MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so timeout 300 python tools/counters.py --config $c ${CTR_ARGS:-} > $O/$c.log 2>&1
  cat $O/$c.log
done
