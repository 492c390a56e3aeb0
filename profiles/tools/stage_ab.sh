#!/bin/bash
# helm_mfma_kernel: metric data per slice from memory (dependent round trips) vs chunks of it requested ahead and parked in LDS
# (CUDDH_HELM_MFMA_STAGE = 100 waves/SIMD + 10 stiffness chunks + mass chunks); parity of each staged form first.
# usage: stage_ab.sh [wide] [variants...]        (wide: also n_basis 5 on the matrix cores and n_basis 8)
WIDE=0; [ "${1:-}" = wide ] && { WIDE=1; shift; }
V=${@:-"0 322 321 332 222"}
one() { python3 profiles/tools/native_apply.py "$@" 2>&1 | grep "native ordering " | tail -1 | sed 's/.*| //'; }
for st in $V; do
  echo "######## CUDDH_HELM_MFMA_STAGE=$st"
  export CUDDH_HELM_MFMA_STAGE=$st
  [ "$st" != 0 ] && python3 -m pytest tests/test_gpu_parity.py -q -k "native_ordering and mfma" 2>&1 | tail -1
  echo "== n_basis 6, irregular r=5"; one 0 6 20 5
  echo "== n_basis 7, irregular r=5"; one 0 7 20 5
  echo "== n_basis 6, 384^2"; one 384 6 20
  echo "== n_basis 7, 384^2"; one 384 7 20
  if [ $WIDE = 1 ]; then
    echo "== n_basis 8, 384^2"; one 384 8 20
    echo "== n_basis 8, irregular r=5"; one 0 8 20 5
    echo "== n_basis 5, 768^2, matrix cores"; CUDDH_HELM_NB5_MFMA=1 one 768 5 20
  fi
done
