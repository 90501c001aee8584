#!/bin/bash
# Round-end evidence, written under gpurun_out/final/ (copy what is to be judged into profiles/):
#   bench_line.json                 default `python bench.py` line (cpu_baseline + fp8 leg included)
#   bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the same command (summary only)
#   bench_phase_breakdown.txt       per-kernel time inside image->first-token and decode (tools/analyze_trace.py)
#   bench_line_under_rocprof.json   the line printed by the profiled run
#   configs.txt                     tools/bench_stage1.py, bench_vcr.py (bf16, fp8), bench_batched.py (bf16, fp8)
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.log || exit 1
echo "bench done"; cut -c1-200 $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_final
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/prof_final.log 2>&1 || exit 1
grep "^{" /tmp/prof_final.log | tail -1 > $OUT/bench_line_under_rocprof.json
cp $(find /tmp/prof_final -name "b_kernel_stats.csv") $OUT/bench_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/analyze_trace.py $(find /tmp/prof_final -name "b_kernel_trace.csv") > $OUT/bench_phase_breakdown.txt
echo "profile done"; head -8 $OUT/bench_phase_breakdown.txt
cd $GRAFT_REPO_ROOT
# every step unbuffered and visible as it goes (a silent step looks hung to the box's watchdog): --line-buffered greps, tee
run() { echo "== $*" | tee -a $OUT/configs.txt; timeout -k 10 420 python -u "$@" 2>&1 | grep --line-buffered -v amdgpu.ids | grep --line-buffered "batch\|stage-1\|tiles ->\|Zephyr forward\|whole forward\|qformer\|step time\|device memory" | tee -a $OUT/configs.txt; }
: > $OUT/configs.txt
run tools/bench_stage1.py
run tools/bench_stage1.py 64 32 train 64
run tools/bench_vcr.py
run tools/bench_vcr.py fp8
run tools/bench_vcr.py fp8mfma
VZ_STREAM_ROWS=16 run tools/bench_vcr.py stream
VZ_STREAM_ROWS=32 run tools/bench_vcr.py stream
run tools/bench_vision.py
run tools/bench_batched.py
run tools/bench_batched.py 32 fp8
echo "all done"; cat $OUT/configs.txt | head -30
