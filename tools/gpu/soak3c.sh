set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_r3c; mkdir -p $O
for S in 91 92 93 94 95 96; do
  timeout 900 python tools/soak.py --seed $S --streams 128 --slabs $((3 + S % 4)) > $O/slabs_$S.log 2>&1; tail -1 $O/slabs_$S.log
done
grep -h MISMATCH $O/*.log | head
