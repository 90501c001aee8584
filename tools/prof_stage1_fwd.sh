#!/bin/bash
# rocprofv3 kernel trace of the Stage-1 FORWARD (tools/bench_stage1.py <samples> <layers>: 64 x 5 tiles -> spliced embeddings -> Zephyr forward + loss),
# summarised per kernel over the last pass; the trace itself is not kept.   bash tools/prof_stage1_fwd.sh 64 32
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s1
rocprofv3 --kernel-trace -d /tmp/prof_s1 -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_stage1.py "${1:-64}" "${2:-32}" > /tmp/s1_prof.log 2>&1
grep -v amdgpu /tmp/s1_prof.log | tail -4
python3 - <<P
import csv, collections, glob
f = glob.glob("/tmp/prof_s1/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass = everything after the last-but-one cross_entropy / loss kernel
idx = [i for i, r in enumerate(rows) if "causal_lm_loss" in r["Kernel_Name"] or "cross_entropy" in r["Kernel_Name"] or "lm_loss" in r["Kernel_Name"]]
sel = rows[idx[-2] + 1: idx[-1] + 1] if len(idx) >= 2 else rows
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
d = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]
    d[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[n][1] += 1
busy = sum(v[0] for v in d.values())
print(f"last pass: {len(sel)} launches, wall {(t1 - t0) / 1e6:.1f} ms, kernel time {busy / 1e6:.1f} ms")
for n, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"{n:52s} {v[0] / 1e6:9.2f} ms {100 * v[0] / busy:5.1f} % {v[1]:6d} launches avg {v[0] / v[1] / 1e3:9.1f} us")
P
