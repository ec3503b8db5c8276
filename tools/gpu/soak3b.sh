set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_r3b; mkdir -p $O
for S in $(seq 51 70); do
  timeout 900 python tools/soak.py --seed $S --streams 192 > $O/wave_flat_$S.log 2>&1; tail -1 $O/wave_flat_$S.log
done
for S in 71 72 73 74 75 76; do
  timeout 900 python tools/soak.py --seed $S --streams 192 --ring > $O/wave_ring_$S.log 2>&1; tail -1 $O/wave_ring_$S.log
done
for S in 81 82 83 84 85 86; do
  timeout 900 python tools/soak.py --seed $S --streams 192 --engine workgroup > $O/wg_flat_$S.log 2>&1; tail -1 $O/wg_flat_$S.log
done
grep -h MISMATCH $O/*.log | head
cat $O/*.log | grep "frames compared" | awk '{f+=$(NF-5); b+=$(NF-2)} END {print "TOTAL frames", f, "mismatching streams", b}'
