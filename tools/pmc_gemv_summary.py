#!/usr/bin/env python
"""profiles/r03_pmc_gemv_fetch.json (what bench.py's roofline.traffic reads) from the decode-chain FETCH_SIZE pass of tools/pmc_fetch.sh:
launch-weighted HBM read bytes of the decode GEMV beside its algorithmic bytes (weights + activations of the launches that ran).
    python tools/pmc_gemv_summary.py gpurun_out/pmc_decode_chain_FETCH_SIZE.json profiles/r03_pmc_gemv_fetch.json"""
import json
import sys

src, dst = sys.argv[1:3]
d = json.load(open(src))
H, I, V, QKV = 4096, 14336, 32000, 6144
alg = {"262144": 2 * I * H * 2, "131072": None, "196608": QKV * H * 2}          # by grid size (threads): gate|up, O / down / lm_head, QKV
rows, by_grid, n_all, b_all = [], {}, 0, 0.0
for k in d["kernels"]:
    if not k["kernel"].startswith("gemv_bf16_kernel"):
        continue
    n, b = k["launches"], k["hbm_read_bytes_per_launch"]
    e = by_grid.setdefault(k["grid"], {"n": 0, "bytes": 0.0})
    e["bytes"] = (e["bytes"] * e["n"] + b * n) / (e["n"] + n)
    e["n"] += n
    n_all += n
    b_all += b * n
fused = [k for k in d["kernels"] if k["kernel"].startswith("attn_o_fused")]
out = {"kernel": "gemv_bf16_kernel<1,2,8,true> (decode weight stream, launch chain; the O projection rides in attn_o_fused_kernel)",
       "counter": "FETCH_SIZE (rocprofv3 --pmc FETCH_SIZE --kernel-trace, own pass, VZ_NO_GRAPH=1, tools/pmc_fetch.sh, round 3, attention|O route)",
       "unit_note": "KiB x 1024 x 2 (gfx950 half-count of wide coalesced streams)",
       "launches": n_all, "hbm_read_bytes_per_launch": b_all / n_all, "by_grid": by_grid}
if fused:
    out["attn_o_fused_kernel"] = {"launches": fused[0]["launches"], "hbm_read_bytes_per_launch": fused[0]["hbm_read_bytes_per_launch"],
                                  "algorithmic": "O weights 33.55 MB + K / V of the context 8.4-8.9 MB"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out)[:600])
