set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3r; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --no-cpu --no-h2d --no-extra --config rtty --steps 2 --warmup 1 > $O/b.json 2>$O/b.err
f=$(find $O/tr -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'demod_wave' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
print(rows[0].keys())
for r in rows[-40:]:
    print(r.get('Queue_Id'), r.get('Stream_Id'), r['Kernel_Name'][:40], r.get('Grid_Size'), r.get('Scratch_Size', r.get('Private_Segment_Size')), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
