#!/bin/bash
# Round-3 PMC passes (each in its own rocprofv3 run, --kernel-trace only; the program itself after `--`):
#   FETCH_SIZE / WRITE_SIZE of the decode GEMV, the persistent decode-token kernel, the prefill tile GEMMs (gemm256 / gemm128) and the
#   17..64-row weight streams (gemm_wide) -> gpurun_out/pmc_*.json (bytes = KiB x 1024, FETCH x 2: gfx950 half-count, MI355X notes section HBM)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
summ() {   # $1 = dir, $2 = out json, $3 = label
python3 - "$1" "$2" "$3" <<'P'
import csv, glob, json, sys, collections
d, out, label = sys.argv[1:4]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
if not f:
    print("no counter file in", d); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70], r["Grid_Size"])
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"what": label, "unit_note": "FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream: bytes = KiB x 1024 x 2 (WRITE_SIZE x 1)", "kernels": []}
for (k, g), cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    e = {"kernel": k, "grid": g}
    for c, v in cs.items():
        e["launches"] = len(v)
        e[c + "_mean_kib"] = sum(v) / len(v)
        e[("hbm_read" if c == "FETCH_SIZE" else "hbm_write") + "_bytes_per_launch"] = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)
    res["kernels"].append(e)
json.dump(res, open(out, "w"), indent=1)
for e in res["kernels"][:12]:
    print({k: (round(v) if isinstance(v, float) else v) for k, v in e.items() if not k.endswith("_kib")})
P
}
mkdir -p $R/gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm1; VZ_NO_GRAPH=1 VZ_REPS=1 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "gemv_bf16_kernel|attn_decode_fused|attn_o_fused" -d /tmp/pm1 -o p --output-format csv -- python3 $R/tools/bench_decode.py "" > /tmp/pm1.log 2>&1
  echo "== decode launch chain $C"; summ /tmp/pm1 $R/gpurun_out/pmc_decode_chain_$C.json "decode launch chain (GEMV + attention|O launch), configs[2], $C"
  [ "$PMC_ONLY" = chain ] && continue
  rm -rf /tmp/pm2; VZ_NO_GRAPH=1 VZ_REPS=1 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "decode_token_kernel" -d /tmp/pm2 -o p --output-format csv -- python3 $R/tools/bench_decode.py "28=1" > /tmp/pm2.log 2>&1
  echo "== persistent decode-token kernel $C"; summ /tmp/pm2 $R/gpurun_out/pmc_decode_token_$C.json "persistent decode-token kernel, configs[2], $C"
  rm -rf /tmp/pm3; rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "gemm" -d /tmp/pm3 -o p --output-format csv -- python3 $R/tools/pmc_gemm.py > /tmp/pm3.log 2>&1
  echo "== prefill GEMMs $C"; summ /tmp/pm3 $R/gpurun_out/pmc_prefill_gemm_$C.json "Zephyr prefill GEMMs at S = 2048 (operand bytes: QKV 16.8 + 50.3 MB in, 25.2 out; O 16.8 + 33.6 in, 16.8 out; gate-up 16.8 + 234.9 in, 58.7 out; down 58.7 + 117.4 in, 16.8 out), $C"
  rm -rf /tmp/pm4; VZ_ROWS=64 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "wide_tiled|gemm_bf16_kernel" -d /tmp/pm4 -o p --output-format csv -- python3 $R/tools/bench_batched.py 32 > /tmp/pm4.log 2>&1
  echo "== 64-row decode step (bf16) $C"; summ /tmp/pm4 $R/gpurun_out/pmc_rows64_$C.json "64-row decode step, bf16 engine: gemm_wide + 128^2 split-K tile route, $C"
done
tail -3 /tmp/pm1.log /tmp/pm2.log
