#!/bin/bash
# rocprofv3 kernel trace of tools/bench_batched.py, summarised per kernel over the widest batch's decode steps (the trace itself is not kept)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_bb -o bb --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_batched.py "$@" > /tmp/bb.log 2>&1
grep batch /tmp/bb.log
python3 - <<P
import csv, collections, glob
f = glob.glob("/tmp/prof_bb/**/bb_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-400 * 40:]
d = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    k = (n, r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"])
    d[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[k][1] += 1
for n, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{n[0]:64s} grid {n[1]:>14s} {v[0] / 1e6:8.2f} ms {v[1]:6d} calls avg {v[0] / v[1] / 1e3:8.1f} us")
P
