#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel time inside the LAST
image->first-token phase (im2col .. first argmax) and inside the decode phase after it."""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")[:60]


starts = [i for i, r in enumerate(rows) if "im2col" in r[2]]
# the last TIMED step = the last request whose decode phase is a full greedy generate (>= 100 argmax tails): the legs bench.py runs
# after the timed region (roofline replays, the 16-token parity request, the sampled streamer path) do not qualify
i0 = None
for k in range(len(starts) - 1, -1, -1):
    end = starts[k + 1] if k + 1 < len(starts) else len(rows)
    if sum(1 for r in rows[starts[k]:end] if "argmax" in r[2]) >= 100:
        i0 = starts[k]
        break
if i0 is None:
    sys.exit("no full greedy request (im2col .. >= 100 argmax launches) in this trace")
first_arg = next(i for i in range(i0, len(rows)) if "argmax" in rows[i][2])
nxt = next((i for i in starts if i > i0), len(rows))
for title, a, b in (("image->first-token", i0, first_arg + 1), ("decode", first_arg + 1, nxt)):
    acc = defaultdict(lambda: [0, 0])
    for s, e, n in rows[a:b]:
        acc[short(n)][0] += e - s
        acc[short(n)][1] += 1
    span = rows[b - 1][1] - rows[a][0]
    busy = sum(v[0] for v in acc.values())
    print(f"== {title}: span {span / 1e6:.3f} ms, kernel-busy {busy / 1e6:.3f} ms, {b - a} launches")
    for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"   {k:60s} {t / 1e6:9.3f} ms  {c:6d} calls  avg {t / c / 1e3:8.2f} us")
