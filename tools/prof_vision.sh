#!/bin/bash
# rocprofv3 kernel summary of tools/bench_vision.py (CLIP tower + Q-Former of the 5-tile workload); stats land in gpurun_out/
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_vision
rocprofv3 --kernel-trace --stats -d /tmp/prof_vision -o v --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_vision.py "$@" > /tmp/prof_vision.log 2>&1 < /dev/null
tail -3 /tmp/prof_vision.log
f=$(find /tmp/prof_vision -name "v_kernel_stats.csv" | head -1)
if [ -z "$f" ]; then echo "no kernel stats written"; find /tmp/prof_vision | head; exit 1; fi
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/vision_kernel_stats.csv
head -24 "$f" | cut -c1-150
