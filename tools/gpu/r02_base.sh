set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_base; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/bench_configs.py > $O/bench_configs.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O -o clk -- python bench.py --steps 5 --warmup 1 --no-cpu > $O/clk.log 2>&1
rocm-smi --showclocks > $O/smi_idle.log 2>&1
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 0.5; rocm-smi --showclocks | grep -E "sclk|mclk|fclk" ; done ) > $O/smi_load.log 2>&1 &
timeout 120 python bench.py --steps 4000 --warmup 5 --no-cpu > $O/bench_long.json 2>&1
wait
tail -3 $O/pytest.log; cat $O/bench_configs.log; cat $O/bench_long.json | cut -c1-400
