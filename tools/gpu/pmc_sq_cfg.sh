# SQ instruction mix / busy counters for one bench config: tools/gpu/pmc_sq_cfg.sh rtty
set -u
C=${1:-rtty}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcsq_$C; mkdir -p $O
B="python bench.py --config $C --no-cpu --steps 3 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O -o sq -- $B > $O/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $O -o sq2 -- $B > $O/sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O -o clk -- $B > $O/clk.log 2>&1
python - "$O" <<'PY'
import csv,glob,collections,sys
for f in sorted(glob.glob(sys.argv[1]+'/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'demod' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        print(k, 'n',len(v), 'mean %.6g'%(sum(v)/len(v)))
for f in sorted(glob.glob(sys.argv[1]+'/clk_kernel_trace.csv')):
    d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'demod' in r['Kernel_Name']]
    print('kernel ns mean', sum(d)/len(d), len(d))
PY
