# last soak of the round, final binary: every variant
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_r3g; mkdir -p $O
run() { n=$1; shift; timeout 900 python tools/soak.py "$@" > $O/$n.log 2>&1; tail -1 $O/$n.log; }
for S in 121 122 123 124 125 126; do run chain_$S --seed $S --streams 160 --chain; done
for S in 131 132 133 134 135 136; do run flat_$S --seed $S --streams 192; done
for S in 141 142 143; do run ring_$S --seed $S --streams 192 --ring; done
for S in 151 152 153; do run wg_$S --seed $S --streams 192 --engine workgroup; done
for S in 161 162; do run lean_$S --seed $S --streams 192 --no-episodes; done
for S in 171 172; do run leanchain_$S --seed $S --streams 160 --no-episodes --chain; done
for S in 181 182; do run slabs_$S --seed $S --streams 128 --slabs 5; done
grep -h MISMATCH $O/*.log | head
python - <<'PY'
import glob, re
tot = 0
for f in glob.glob('gpurun_out/soak_r3g/*.log'):
    m = re.search(r'(\d+) frames compared, (\d+) mismatching', open(f).read())
    if m: tot += int(m.group(1)); assert m.group(2) == '0', f
print('total frames', tot)
PY
