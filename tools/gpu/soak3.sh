set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_r3; mkdir -p $O
for S in 41 42 43; do
  timeout 900 python tools/soak.py --seed $S --streams 192 > $O/wave_flat_$S.log 2>&1; tail -1 $O/wave_flat_$S.log
done
timeout 900 python tools/soak.py --seed 44 --streams 192 --ring > $O/wave_ring_44.log 2>&1; tail -1 $O/wave_ring_44.log
timeout 900 python tools/soak.py --seed 45 --streams 192 --engine workgroup > $O/wg_flat_45.log 2>&1; tail -1 $O/wg_flat_45.log
grep -h MISMATCH $O/*.log | head
grep -h "rtty\|^50 " $O/wave_flat_41.log
