#!/bin/bash
# bench.py under rocprofv3 --kernel-trace --stats (the per-kernel summary committed under profiles/), then unprofiled.
# (Round 2 passed --no-extra to the profiled run because the process died with SIGSEGV inside the profiler's frames during
# the fp32 sub-record.  Round 3 traced it with tools/rocprof_crash_probe.py to the host running 100 graph launches ahead
# of the GPU under the profiler - nothing to do with plan teardown - and the sampler now bounds its run-ahead, so the
# profiled run is the whole bench again.)
set -e
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 > $OUT/bench.json.log 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o b -- python3 bench.py --steps 1 --warmup 1 --cpu-calls 0 > $OUT/bench_profiled.json.log 2> $OUT/rocprof.err
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/bench.json.log
