set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_full; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/gpu.log 2>&1; echo "rc=$?" >> $O/gpu.log
tail -3 $O/gpu.log
bash tools/profile_round.sh r02 "1200 rtty 12000 same" 2>&1 | tail -30
