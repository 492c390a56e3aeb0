#!/bin/bash
# same-box A/B of helm_mfma_kernel build variants at n_basis 6 (profiles/tools/build_variant.py): nb6_ab.sh VARIANT...
for v in "$@"; do
  echo "== variant $v"
  if [ $v = base ]; then L="X=1"; else L="CUDDH_AMD_LIBRARY_VARIANT=libcuddh_amd_$v.so"; fi
  env $L python3 profiles/tools/native_apply.py 0 6 20 5 2>&1 | grep "ordering" | tail -1 | cut -c1-230 | sed "s/^/irregular r=5 nb=6: /"
  env $L python3 profiles/tools/native_apply.py 384 6 20 2>&1 | grep "ordering" | tail -1 | cut -c1-230 | sed "s/^/384^2 nb=6: /"
done
