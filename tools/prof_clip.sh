#!/bin/bash
# rocprofv3 kernel trace of the CLIP tower + fusion at T tiles (default 5), summarised per (kernel, grid) over the last 10 passes
cd /tmp && export TMPDIR=/tmp
T=${1:-5}
cat > /tmp/clip_only.py <<P
import os, sys
sys.path.insert(0, os.path.join("$GRAFT_REPO_ROOT", "vision-zephyr_amd"))
import torch
from vz_hip.engine import Engine
from vz_hip.synth import ArchConfig
cfg = ArchConfig(n_layers=1)
eng = Engine(cfg, max_batch=1, max_ctx=64, max_tiles=max(8, $T))
eng.load_synthetic(0)
tiles = torch.randn($T, 3, 336, 336).cuda().bfloat16()
for _ in range(12):
    eng.clip_fused_features(tiles)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    eng.clip_fused_features(tiles)
e.record(); torch.cuda.synchronize()
print(f"clip tower + fusion at $T tiles: {s.elapsed_time(e) / 10:.3f} ms")
P
rm -rf /tmp/prof_clip
rocprofv3 --kernel-trace -d /tmp/prof_clip -o c --output-format csv -- python3 /tmp/clip_only.py > /tmp/clip.log 2>&1
tail -1 /tmp/clip.log
python3 - <<P
import csv, collections, glob
f = glob.glob("/tmp/prof_clip/**/c_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_pass = 10
per = len([r for r in rows if "im2col" in r["Kernel_Name"]])
start = [i for i, r in enumerate(rows) if "im2col" in r["Kernel_Name"]][-n_pass]
sel = rows[start:]
d = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:56]
    k = (n, r["Grid_Size_X"] + "x" + r["Grid_Size_Y"])
    d[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[k][1] += 1
tot = sum(v[0] for v in d.values())
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print(f"per pass: kernel-busy {tot / n_pass / 1e6:.3f} ms, span {span / n_pass / 1e6:.3f} ms")
for n, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"{n[0]:56s} grid {n[1]:>12s} {v[0] / n_pass / 1e3:8.1f} us/pass {v[1] // n_pass:4d} calls avg {v[0] / v[1] / 1e3:7.1f} us")
P
