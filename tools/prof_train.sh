#!/bin/bash
# rocprofv3 kernel trace of the Stage-1 optimiser step (tools/bench_stage1.py <samples> <layers> train <mb>), summarised per kernel over
# the LAST step; the trace itself is not kept.   bash tools/prof_train.sh 64 32 16
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_train
rocprofv3 --kernel-trace -d /tmp/prof_train -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_stage1.py "${1:-64}" "${2:-32}" train "${3:-16}" > /tmp/train_prof.log 2>&1
grep -v amdgpu /tmp/train_prof.log | tail -4
python3 - <<P
import csv, collections, glob
f = glob.glob("/tmp/prof_train/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last optimiser step = everything after the last-but-one run of adamw_kernel launches
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
ends = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] != i + 1]
sel = rows[ends[-2] + 1: ends[-1] + 1] if len(ends) >= 2 else rows
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
d = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    d[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[n][1] += 1
busy = sum(v[0] for v in d.values())
print(f"last step: {len(sel)} launches, wall {(t1 - t0) / 1e6:.1f} ms, kernel time {busy / 1e6:.1f} ms")
for n, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"{n:48s} {v[0] / 1e6:9.2f} ms {100 * v[0] / busy:5.1f} % {v[1]:6d} launches avg {v[0] / v[1] / 1e3:9.1f} us")
P
