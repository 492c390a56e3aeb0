#!/bin/bash
# BASELINE config 5's kernels: fused apply at n_basis 5 (768^2 structured, general layout), 6 and 7 (the reference's unstructured
# fixture refined 5 times: 121,856 quads), reference vs plan-native ordering, mass weights through registers or LDS-DMA.
# usage: config5_ab.sh [quick]
run() { echo "== $*"; env "${@:1:$#-1}" python3 profiles/tools/native_apply.py ${!#} 2>&1 | grep -v amdgpu.ids | tail -n +1; }
for dma in 0 1; do
  echo "######## CUDDH_HELM_MFMA_DMA=$dma"
  export CUDDH_HELM_MFMA_DMA=$dma
  echo "== n_basis 6, irregular r=5"; python3 profiles/tools/native_apply.py 0 6 20 5 2>&1 | grep -v amdgpu.ids
  echo "== n_basis 7, irregular r=5"; python3 profiles/tools/native_apply.py 0 7 20 5 2>&1 | grep -v amdgpu.ids
  echo "== n_basis 6, 384^2"; python3 profiles/tools/native_apply.py 384 6 20 2>&1 | grep -v amdgpu.ids
  echo "== n_basis 7, 384^2"; python3 profiles/tools/native_apply.py 384 7 20 2>&1 | grep -v amdgpu.ids
  echo "== n_basis 8, 384^2"; python3 profiles/tools/native_apply.py 384 8 20 2>&1 | grep -v amdgpu.ids
  echo "== n_basis 5, 768^2, matrix cores"; CUDDH_HELM_NB5_MFMA=1 python3 profiles/tools/native_apply.py 768 5 20 2>&1 | grep -v amdgpu.ids
done
unset CUDDH_HELM_MFMA_DMA
echo "== n_basis 5, 768^2, one element per lane (helm_patch_kernel)"; python3 profiles/tools/native_apply.py 768 5 20 2>&1 | grep -v amdgpu.ids
