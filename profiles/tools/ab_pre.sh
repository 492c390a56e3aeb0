python -m pytest tests/test_gpu_parity.py -x -q -k "patch_sizes or fused_helmholtz or fused_apply" 2>&1 | tail -3
for i in 1 2; do CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 1024 4 2>&1 | grep -i "fused" | head -4; done
