#!/bin/bash
# Dispatch timeline of one rank's share of a DDH action with the split (overlap) schedule: do the boundary and interior
# launches run concurrently?   writes gpurun_out/overlap_timeline.txt
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_ov
rocprofv3 --kernel-trace -d gpurun_out/prof_ov -- python3 profiles/tools/shard_overlap.py 1024 8 3 > gpurun_out/overlap_run.txt 2>&1
DB=$(find gpurun_out/prof_ov -name "*_results.db" | head -1)
cat gpurun_out/overlap_run.txt | grep "overlap=" > gpurun_out/overlap_timeline.txt
python3 profiles/tools/kernel_timeline_from_db.py "$DB" ddh_mfma 24 >> gpurun_out/overlap_timeline.txt
rm -rf gpurun_out/prof_ov
cat gpurun_out/overlap_timeline.txt
