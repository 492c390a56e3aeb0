#!/bin/bash
# rocprofv3 --kernel-trace --stats of config 5's fused applies on the round's final build (reference + native ordering of each):
# n_basis 5 768^2, n_basis 6 / 7 / 8 on the irregular 121,856-quad mesh.   writes gpurun_out/r03/config5_kernel_stats.csv
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
export CUDDH_PLAN_AFFINE=0
OUT=gpurun_out/r03/config5_kernel_stats.csv
: > $OUT
for c in "768 5 20" "0 6 20 5" "0 7 20 5" "0 8 20 5"; do
  rm -rf gpurun_out/prof_c5
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -- python3 profiles/tools/native_apply.py $c > gpurun_out/prof_c5.log 2>&1
  DB=$(find gpurun_out/prof_c5 -name "*_results.db" | head -1)
  python3 profiles/tools/kernel_stats_from_db.py "$DB" gpurun_out/prof_c5.csv > /dev/null
  echo "# native_apply.py $c : $(grep elements gpurun_out/prof_c5.log | head -1)" >> $OUT
  echo "# event-timed in the same (profiled) process: $(grep "native ordering " gpurun_out/prof_c5.log | tail -1)" >> $OUT
  grep "helm_\|\"Name\"" gpurun_out/prof_c5.csv >> $OUT
  rm -rf gpurun_out/prof_c5
done
cat $OUT | cut -c1-220
