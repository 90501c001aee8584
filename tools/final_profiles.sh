#!/bin/bash
# Round-end evidence, written under gpurun_out/final/ (copy what is to be judged into profiles/):
#   bench_line.json                 default `python bench.py` line (cpu_baseline + fp8 leg included)
#   bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the same command (summary only)
#   bench_phase_breakdown.txt       per-kernel time inside image->first-token and decode (tools/analyze_trace.py)
#   bench_line_under_rocprof.json   the line printed by the profiled run
#   configs.txt                     tools/bench_stage1.py, bench_vcr.py (bf16, fp8), bench_batched.py (bf16, fp8)
#   fused_stamps.txt                role timeline of the opt-in one-launch attention half
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.log || exit 1
echo "bench done"; cut -c1-200 $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_final
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/prof_final.log 2>&1 || exit 1
tail -1 /tmp/prof_final.log > $OUT/bench_line_under_rocprof.json
cp $(find /tmp/prof_final -name "b_kernel_stats.csv") $OUT/bench_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/analyze_trace.py $(find /tmp/prof_final -name "b_kernel_trace.csv") > $OUT/bench_phase_breakdown.txt
echo "profile done"; head -8 $OUT/bench_phase_breakdown.txt
cd $GRAFT_REPO_ROOT
{
  echo "== tools/bench_stage1.py"; timeout -k 10 300 python tools/bench_stage1.py 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== tools/bench_vcr.py"; timeout -k 10 300 python tools/bench_vcr.py 2>&1 | grep batch
  echo "== tools/bench_vcr.py fp8"; timeout -k 10 300 python tools/bench_vcr.py fp8 2>&1 | grep batch
  echo "== tools/bench_batched.py"; timeout -k 10 300 python tools/bench_batched.py 2>&1 | grep batch
  echo "== tools/bench_batched.py 32 fp8"; timeout -k 10 300 python tools/bench_batched.py 32 fp8 2>&1 | grep batch
} > $OUT/configs.txt
timeout -k 10 100 python tools/fused_stamps.py 2048 8 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/fused_stamps.txt
echo "all done"; cat $OUT/configs.txt | head -30
