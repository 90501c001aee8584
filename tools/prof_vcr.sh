#!/bin/bash
# rocprofv3 kernel trace of the config-5 miniature at one batch size (tools/bench_vcr.py [fp8|fp8mfma], VZ_ROWS=<rows>), summarised per kernel over the
# whole run (3 batches: 1 warm-up + 2 timed); the trace itself is not kept.     VZ_ROWS=64 bash tools/prof_vcr.sh fp8mfma
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_vcr
rocprofv3 --kernel-trace -d /tmp/prof_vcr -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_vcr.py "$@" > /tmp/vcr_prof.log 2>&1
grep batch /tmp/vcr_prof.log | cut -c1-150
python3 - <<P
import csv, collections, glob
f = glob.glob("/tmp/prof_vcr/**/t_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]
    d[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[n][1] += 1
busy = sum(v[0] for v in d.values())
print(f"kernel time {busy / 1e6:.1f} ms over the whole run")
for n, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:26]:
    print(f"{n:56s} {v[0] / 1e6:9.2f} ms {100 * v[0] / busy:5.1f} % {v[1]:7d} launches avg {v[0] / v[1] / 1e3:9.1f} us")
P
