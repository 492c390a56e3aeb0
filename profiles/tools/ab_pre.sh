python -m pytest tests/test_gpu_parity.py -x -q -k "patch_sizes or fused" 2>&1 | tail -2
for w in 1 0 1 0 1 0; do echo "== WIDE=$w"; CUDDH_HELM_WIDE=$w CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 1024 4 2>&1 | grep "fused"; done
