#!/bin/bash
# ONE profiled bench run (rocprofv3 --kernel-trace --stats, all sub-records on) with the given plan-teardown mode:
#   drop    = round 2's form (del the plans, empty_cache) - the run that died with SIGSEGV inside the profiler's frames
#   keep    = every sampler / arena stays alive
#   ordered = device idle -> captured graphs destroyed -> plans dropped -> empty_cache (the shipped form)
# usage: tools/run_rocprof_teardown.sh MODE [OUTDIR]      (one GPU step per call: a crash ends the call)
export TMPDIR=/tmp
MODE=${1:-ordered}
OUT=${2:-gpurun_out/teardown_$MODE}
rm -rf $OUT; mkdir -p $OUT
ADK_BENCH_PHASES=1 timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o b -- python3 bench.py --steps 1 --warmup 1 --cpu-calls 0 --teardown $MODE > $OUT/bench.json.log 2> $OUT/rocprof.err
rc=$?
find $OUT -name "*kernel_trace.csv" -delete
echo "mode=$MODE rc=$rc" | tee $OUT/result.txt
grep -c "SIGSEGV" $OUT/rocprof.err | sed 's/^/SIGSEGV lines: /' | tee -a $OUT/result.txt
tail -c 300 $OUT/bench.json.log
exit 0
