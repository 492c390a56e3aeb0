#!/bin/bash
# same-box A/B of helm_mfma_kernel build variants (profiles/tools/build_variant.py): nb6_ab.sh VARIANT...
for v in "$@"; do
  echo "== variant $v"
  if [ $v = base ]; then L="X=1"; else L="CUDDH_AMD_LIBRARY_VARIANT=libcuddh_amd_$v.so"; fi
  for nb in 6 7; do env $L python3 profiles/tools/native_apply.py 0 $nb 20 5 2>&1 | grep "ordering" | tail -1 | cut -c1-230 | sed "s/^/irregular r=5 nb=$nb: /"; done
  for nb in 6 7 8; do env $L python3 profiles/tools/native_apply.py 384 $nb 20 2>&1 | grep "ordering" | tail -1 | cut -c1-230 | sed "s/^/384^2 nb=$nb: /"; done
done
