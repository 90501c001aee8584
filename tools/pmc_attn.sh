#!/bin/bash
# PMC passes over the prefill attention kernel (own passes, --kernel-trace only): bash tools/pmc_attn.sh
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_WAVES"; do
  rm -rf /tmp/pmc_attn
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "flash_attn2_kernel<128, false, true>" -d /tmp/pmc_attn -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py attn > /tmp/pmc_attn.log 2>&1
  python3 - <<P
import csv, glob, collections
f = glob.glob("/tmp/pmc_attn/**/p_counter_collection.csv", recursive=True)
if not f:
    print("no counter file for: $set"); raise SystemExit
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    v = sorted(v)
    print(f"{k:28s} median {v[len(v) // 2]:14.0f}  (n={len(v)})")
P
done
