cd /tmp && export TMPDIR=/tmp
for t in "" "30=0"; do
rm -rf /tmp/pd; VZ_REPS=1 rocprofv3 --kernel-trace -d /tmp/pd -o d --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_decode.py "$t" > /tmp/pd.log 2>&1
grep -i "tok/s\|tokens" /tmp/pd.log | tail -2
python3 - <<P
import csv, collections, glob
f = glob.glob("/tmp/pd/**/d_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-161 * 60:]
d = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50]
    k = (n, r["Grid_Size_X"])
    d[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[k][1] += 1
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"]); busy = sum(v[0] for v in d.values())
print(f"setting '$t': span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms over {len(sel)} launches")
for n, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:7]:
    print(f"  {n[0]:50s} grid {n[1]:>8s} {v[1]:6d} calls avg {v[0] / v[1] / 1e3:7.2f} us")
P
done
