set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak; mkdir -p $O
for S in ${SEEDS:-31 32 33 34 35 36}; do
  timeout 600 python tools/soak.py --seed $S --streams 256 > $O/wave_flat_$S.log 2>&1; tail -1 $O/wave_flat_$S.log
  timeout 600 python tools/soak.py --seed $S --streams 256 --ring > $O/wave_ring_$S.log 2>&1; tail -1 $O/wave_ring_$S.log
  timeout 600 python tools/soak.py --seed $S --streams 256 --engine workgroup > $O/wg_flat_$S.log 2>&1; tail -1 $O/wg_flat_$S.log
done
grep -h MISMATCH $O/*.log | head
