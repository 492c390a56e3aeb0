python -m pytest tests/test_gpu_parity.py -x -q -k "fused_helmholtz or config5 or tiny" 2>&1 | tail -2
for nb in 6 7; do python profiles/tools/unstructured_apply.py 5 $nb 2>&1 | grep "fused"; done
for nb in 6 7 8; do CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 384 $nb 2>&1 | grep "fused"; done
