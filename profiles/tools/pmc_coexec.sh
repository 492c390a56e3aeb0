#!/bin/bash
# Do the matrix pipe and the VALU of a SIMD work at the same time in ddh_mfma_kernel (DDH kernel 5)?
# Separate rocprofv3 --pmc passes (--kernel-trace only, program directly after `--`), 256^2 elements = 4,096 subdomains, nt = 5120.
# usage: pmc_coexec.sh TAG [env assignments for the run, e.g. CUDDH_DDH_STAGGER=1]     writes gpurun_out/pmc_coexec_TAG.txt
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_coexec_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
for pass in "mfma:SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA" \
            "valu:SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
            "misc:SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rm -rf $OUT/$name
  rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$name -- python3 profiles/tools/run_kernel.py ddh 256 2 5 > $OUT/$name.log 2>&1
  python3 profiles/tools/pmc_summary.py $OUT/$name > $OUT/$name.txt 2>&1
  rm -rf $OUT/$name
done
{ echo "ddh_mfma_kernel, 256^2 elements (4,096 subdomains), n_basis 4, nt = 5120, $*"; cat $OUT/mfma.txt $OUT/valu.txt $OUT/misc.txt; } > gpurun_out/pmc_coexec_$TAG.txt
cat gpurun_out/pmc_coexec_$TAG.txt
