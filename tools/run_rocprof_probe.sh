#!/bin/bash
# one profiled run of tools/rocprof_crash_probe.py; arguments after the output directory go to the probe
export TMPDIR=/tmp
OUT=$1; shift
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p -- python3 tools/rocprof_crash_probe.py "$@" > $OUT/out.log 2> $OUT/err.log
rc=$?
find $OUT -name "*kernel_trace.csv" -delete
echo "probe $* rc=$rc" | tee $OUT/result.txt
grep "^\[probe" $OUT/err.log | tee -a $OUT/result.txt
exit 0
