#!/bin/bash
# SQ instruction-mix counters for the bench kernel (run through gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_sq; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d $O -o sq -- python bench.py --steps 5 --warmup 1 --no-cpu > $O/sq.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY \
    --output-format csv -d $O -o sq2 -- python bench.py --steps 5 --warmup 1 --no-cpu > $O/sq2.log 2>&1
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmc_sq/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'demod' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        print(k, 'n',len(v), 'mean %.5g'%(sum(v)/len(v)))
PY
