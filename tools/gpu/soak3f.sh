# soak of the lean replay (no episode records asked for), final kernels
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_r3f; mkdir -p $O
for S in 101 102 103; do
  timeout 900 python tools/soak.py --seed $S --streams 192 --no-episodes > $O/lean_$S.log 2>&1; tail -1 $O/lean_$S.log
done
for S in 104 105; do
  timeout 900 python tools/soak.py --seed $S --streams 160 --no-episodes --chain > $O/lean_chain_$S.log 2>&1; tail -1 $O/lean_chain_$S.log
done
for S in 106 107; do
  timeout 900 python tools/soak.py --seed $S --streams 192 --no-episodes --engine workgroup > $O/lean_wg_$S.log 2>&1; tail -1 $O/lean_wg_$S.log
done
timeout 900 python tools/soak.py --seed 108 --streams 192 > $O/full_108.log 2>&1; tail -1 $O/full_108.log
grep -h MISMATCH $O/*.log | head
