#!/bin/bash
# Is helm_mfma_kernel bound by instruction issue?  VALU / matrix-pipe busy cycles against the kernel's busy cycles, separate
# rocprofv3 --pmc passes (--kernel-trace only, program directly after `--`).   usage: pmc_mfma_busy.sh TAG NX NB [REFINE] [env assignments]
set -u
TAG=$1; NX=$2; NB=$3; RF=${4:--1}; shift 4 2>/dev/null || shift $#
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_busy_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CYCLES" \
            "valu:SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
            "misc:SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rm -rf $OUT/$name
  rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$name -- python3 profiles/tools/run_kernel.py helmn $NX 5 0 $NB $RF > $OUT/$name.log 2>&1
  python3 profiles/tools/pmc_summary.py $OUT/$name > $OUT/$name.txt 2>&1
  rm -rf $OUT/$name
done
{ echo "fused apply on plan-native vectors, nx $NX, n_basis $NB, refine $RF, $*"; grep "kernel:" $OUT/mfma.log; cat $OUT/mfma.txt $OUT/valu.txt $OUT/misc.txt; } > gpurun_out/pmc_busy_$TAG.txt
cat gpurun_out/pmc_busy_$TAG.txt
