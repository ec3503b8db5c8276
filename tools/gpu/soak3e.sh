# soak on the round's final kernels (after the ring / auto-carrier instantiations were split off)
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_r3e; mkdir -p $O
for S in 81 82 83 84 85 86; do
  timeout 900 python tools/soak.py --seed $S --streams 160 --chain > $O/chain_$S.log 2>&1; tail -1 $O/chain_$S.log
done
for S in 91 92 93 94; do
  timeout 900 python tools/soak.py --seed $S --streams 192 > $O/wave_flat_$S.log 2>&1; tail -1 $O/wave_flat_$S.log
done
for S in 95 96; do
  timeout 900 python tools/soak.py --seed $S --streams 192 --ring > $O/wave_ring_$S.log 2>&1; tail -1 $O/wave_ring_$S.log
done
for S in 97 98; do
  timeout 900 python tools/soak.py --seed $S --streams 192 --engine workgroup > $O/wg_$S.log 2>&1; tail -1 $O/wg_$S.log
done
timeout 900 python tools/soak.py --seed 99 --streams 128 --slabs 4 > $O/slabs_99.log 2>&1; tail -1 $O/slabs_99.log
grep -h MISMATCH $O/*.log | head
