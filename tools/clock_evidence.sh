#!/bin/bash
# Round-3 re-measurement of the prefill GEMMs' in-kernel clock (s_memtime / s_memrealtime stamps of the diagnostic build, after >= 1 s of
# back-to-back launches) on random and on all-zero operands -> gpurun_out/clock_evidence.txt
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clock_evidence.txt
: > $OUT
for shape in "2048 28672 4096 1" "2048 4096 14336 1" "2048 6144 4096 0" "2048 4096 4096 0" "4096 4096 4096 0"; do
  echo "== random operands" >> $OUT
  python tools/gemm_stamps.py $shape 2>/dev/null | head -4 >> $OUT
  echo "== all-zero operands" >> $OUT
  VZ_ZERO=1 python tools/gemm_stamps.py $shape 2>/dev/null | head -2 >> $OUT
done
cat $OUT
