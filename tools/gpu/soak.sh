# tools/soak.py over SEEDS (default six) x 256 streams x every variant: one launch (both engines, RING
# addressing), chained (both engines), slabs (both engines, RING slabs), without episode records
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak; mkdir -p $O
run() { tag=$1; shift; timeout -s KILL 280 python tools/soak.py --seed $S --streams ${STREAMS:-256} "$@" > $O/${tag}_$S.log 2>&1; tail -1 $O/${tag}_$S.log; }
for S in ${SEEDS:-31 32 33 34 35 36}; do
  run wave_flat; run wave_ring --ring; run wg_flat --engine workgroup
  run wave_chain --chain; run wg_chain --chain --engine workgroup
  run wave_slabs --slabs 5; run wg_slabs --slabs 5 --engine workgroup; run ring_slabs --slabs 4 --ring
  run noeps --no-episodes
done
grep -h MISMATCH $O/*.log | head
