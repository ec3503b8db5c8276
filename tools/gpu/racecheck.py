"""Does tests/test_gpu_streams.py see the race it was written for?  Its script with the wrapper's
output fills back on torch's current stream (the behaviour before round 5's fix): prints RACE_SEEN
and the first differing (mode, repetition, lane, field).  GPU_MAX_HW_QUEUES=4|8 MIFSK_ROOT=$PWD python tools/gpu/racecheck.py"""
import os, sys, contextlib
sys.path.insert(0, os.environ["MIFSK_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MIFSK_ROOT"], "tests"))
import minimodem_amd as M
M._on = lambda torch, stream: contextlib.nullcontext()      # the round-4 behaviour: fills on torch's current stream
import test_gpu_streams as T
try:
    exec(T.SCRIPT)
except AssertionError as e:
    print("RACE_SEEN", e)
