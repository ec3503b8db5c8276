cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_mem; mkdir -p $O
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_32B_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCC_BUSY_avr TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o k$i -- python bench.py --steps 3 --warmup 1 --no-cpu > $O/k$i.log 2>&1
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o u$i -- tools/ubench/stream_pattern 0 > $O/u$i.log 2>&1
done
ls $O | head -40
