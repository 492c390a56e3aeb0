#!/bin/bash
# helm_lane_kernel on native vectors: straight-line gather with the scalars through the scalar cache (base) vs the gather loop
# (slowgather = -DHELM_LANE_FAST_GATHER=0), same box, alternating
python3 -m pytest tests/test_gpu_parity.py -q -k "native_ordering" 2>&1 | tail -1
one() { CUDDH_PLAN_AFFINE=0 python3 profiles/tools/native_apply.py "$@" 2>&1 | grep "native ordering " | tail -1 | sed 's/.*| //'; }
for v in base slowgather base slowgather; do
  if [ $v = base ]; then unset CUDDH_AMD_LIBRARY_VARIANT; else export CUDDH_AMD_LIBRARY_VARIANT=libcuddh_amd_$v.so; fi
  echo "######## $v"
  echo "n_basis 4, 1024^2: $(one 1024 4 30)"
  echo "n_basis 3, 1024^2: $(one 1024 3 30)"
  echo "n_basis 2, 1024^2: $(one 1024 2 30)"
  echo "n_basis 4, irregular r=6 (487k quads): $(one 0 4 30 6)"
done
