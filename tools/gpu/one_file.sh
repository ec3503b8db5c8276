#!/bin/bash
# What a batch of ONE costs: one 10-second Bell-202 file through the reference's
# main() -- unpatched over the five legacy symbols, with the rx-batch patch (RING addressing, the
# default, and --flat is not an option of the reference's main: MIFSK_CLI_FLAT is the shim's), and
# the reference on the CPU -- with the batch path's own phase timing.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import minimodem_amd as M, bench
cfg = M.rx_config("1200")
x, p = bench.make_stream(M, cfg, 0)
full = np.zeros(480000, np.float32); full[:len(x)] = x
bench.write_wav_f32("/tmp/one.wav", full, 48000)
PY
for exe in minimodem_ref minimodem_mifsk minimodem_mifsk_rxbatch minimodem_mifsk_batch; do
  for i in 1 2 3; do
    S=$(date +%s.%N)
    MIFSK_CLI_TIMING=1 oracle/_ref/$exe --rx --quiet --file /tmp/one.wav 1200 > /tmp/one.out 2> /tmp/one.err
    E=$(date +%s.%N)
    echo "$exe run $i: $(python -c "print('%.1f ms' % (1e3*($E-$S)))") $(wc -c < /tmp/one.out) bytes; $(grep TIMING /tmp/one.err | tr '\n' ' ')"
  done
done
