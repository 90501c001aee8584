#!/bin/bash
# GPU-box CI helper: run pytest files one after another, stop at the first one that was killed
# (timeout / signal) so a hung kernel is never followed by more GPU work.  Logs under gpurun_out/.
mkdir -p gpurun_out
rm -f gpurun_out/parity_metrics.jsonl
for f in "$@"; do
    name=$(basename "$f" .py)
    echo "=== $f" | tee -a gpurun_out/ci.log
    timeout -k 10 ${VZ_CI_TIMEOUT:-600} python -m pytest "$f" -m gpu -q -rfE --no-header -p no:cacheprovider > "gpurun_out/$name.log" 2>&1
    rc=$?
    tail -n 25 "gpurun_out/$name.log"
    echo "rc=$rc" | tee -a gpurun_out/ci.log
    if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
done
exit 0
