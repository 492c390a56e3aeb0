#!/bin/bash
# HBM traffic and wait/occupancy counters of one kernel case, collected as MI355X_MICROARCH.md prescribes: separate
# rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), --kernel-trace only, program directly after `--`.
# usage: pmc_case.sh TAG  run_kernel.py-arguments...        writes gpurun_out/pmc_TAG/{fetch,write,sq}.txt
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "lds:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rm -rf $OUT/$name
  rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$name -- python3 profiles/tools/run_kernel.py "$@" > $OUT/$name.log 2>&1
  python3 profiles/tools/pmc_summary.py $OUT/$name > $OUT/$name.txt 2>&1
  rm -rf $OUT/$name
done
grep "kernel:\|elements" $OUT/fetch.log > $OUT/case.txt
cat $OUT/case.txt $OUT/fetch.txt $OUT/write.txt $OUT/sq.txt $OUT/lds.txt
