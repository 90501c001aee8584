#!/bin/bash
# rocprofv3 kernel trace of bench.py (one timed step), summarised with tools/analyze_trace.py; the trace itself is not kept.
#   bash tools/prof_bench.sh [bench.py flags...]      (VZ_TUNE is passed through)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace -d /tmp/prof_bench -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp8-leg --steps 1 --warmup 1 "$@" > /tmp/bench_prof.log 2>&1
grep "^{" /tmp/bench_prof.log | tail -1 | cut -c1-300
python3 $GRAFT_REPO_ROOT/tools/analyze_trace.py $(find /tmp/prof_bench -name "b_kernel_trace.csv") 
