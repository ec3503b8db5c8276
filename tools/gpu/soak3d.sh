# soak on the round's final kernels: chained launches forced onto the small batches, plus plain
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_r3d; mkdir -p $O
for S in 61 62 63 64 65 66; do
  timeout 900 python tools/soak.py --seed $S --streams 160 --chain > $O/chain_$S.log 2>&1; tail -1 $O/chain_$S.log
done
for S in 71 72 73; do
  timeout 900 python tools/soak.py --seed $S --streams 192 > $O/wave_flat_$S.log 2>&1; tail -1 $O/wave_flat_$S.log
done
timeout 900 python tools/soak.py --seed 74 --streams 192 --ring > $O/wave_ring_74.log 2>&1; tail -1 $O/wave_ring_74.log
timeout 900 python tools/soak.py --seed 75 --streams 128 --slabs 5 > $O/slabs_75.log 2>&1; tail -1 $O/slabs_75.log
grep -h MISMATCH $O/*.log | head
grep -h "cut" $O/chain_61.log
