for lib in libcuddh_amd_g11.so libcuddh_amd.so libcuddh_amd_g24w3.so libcuddh_amd_g46w2.so libcuddh_amd_g11.so libcuddh_amd.so; do
  echo "=== $lib"
  export CUDDH_AMD_LIBRARY_VARIANT=$lib
  for nb in 6 7; do python profiles/tools/unstructured_apply.py 5 $nb 2>&1 | grep -i "fused"; done
  for c in "384 6" "384 7" "384 8"; do CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py $c 2>&1 | grep "fused complex"; done
done
unset CUDDH_AMD_LIBRARY_VARIANT
python -m pytest tests/test_gpu_parity.py -x -q -k "fused or patch_sizes" 2>&1 | tail -2
for c in "256 3" "512 3"; do CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py $c 2>&1 | grep "fused complex"; done
