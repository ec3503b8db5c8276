#!/bin/bash
# Kernel time vs. batch size on one GPU (diagnostic): tools/sweep.sh "256 512 1024"
for n in ${1:-256 512 768 1024 1536 2048}; do
  timeout 200 python bench.py --no-cpu --streams $n --steps 10 --warmup 2 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('streams %5d  kernel_ms avg %.4f min %.4f  frac %.3f' % ($n, r['kernel_ms_avg'], r['kernel_ms_min'], r['frac']))"
done
