set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_soak; mkdir -p $O
for S in 21 22; do
  timeout 500 python tools/soak.py --seed $S --streams 192 > $O/wave_flat_$S.log 2>&1; tail -1 $O/wave_flat_$S.log
  timeout 500 python tools/soak.py --seed $S --streams 192 --ring > $O/wave_ring_$S.log 2>&1; tail -1 $O/wave_ring_$S.log
  timeout 500 python tools/soak.py --seed $S --streams 192 --engine workgroup > $O/wg_flat_$S.log 2>&1; tail -1 $O/wg_flat_$S.log
done
grep -h MISMATCH $O/*.log | head
