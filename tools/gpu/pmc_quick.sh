# quick memory counters for one bench config: tools/gpu/pmc_quick.sh rtty
set -u
C=${1:-rtty}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcq_$C; mkdir -p $O
B="python bench.py --config $C --no-cpu --steps 3 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O -o tcc -- $B > $O/tcc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O -o ea -- $B > $O/ea.log 2>&1
python - "$O" <<'PY'
import csv,glob,collections,sys
for f in sorted(glob.glob(sys.argv[1]+'/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'demod' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        print(k, 'n',len(v), 'mean %.6g'%(sum(v)/len(v)))
PY
