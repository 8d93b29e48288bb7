#!/bin/bash
# bench.py under rocprofv3 --kernel-trace --stats (the per-kernel summary committed under profiles/), then unprofiled.
# The profiled run passes --no-extra (same timed region; only the accuracy / fp32 / bf16 sub-records that follow it are
# skipped): with them, rocprofv3 7.2 itself segfaults inside a torch device-to-device copy after the arenas are freed.
set -e
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 > $OUT/bench.json.log 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o b -- python3 bench.py --steps 1 --warmup 1 --cpu-calls 0 --no-extra > $OUT/bench_profiled.json.log 2> $OUT/rocprof.err
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/bench.json.log
