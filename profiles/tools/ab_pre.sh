python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -2
CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 1024 4 2>&1 | grep -v "action(c\|amdgpu" | head -5
for nb in 6 7; do python profiles/tools/unstructured_apply.py 5 $nb 2>&1 | grep "fused\|stiffness"; done
for nb in 5 6 7 8; do CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 384 $nb 2>&1 | grep -v "action(c\|amdgpu\|unfused" | head -4; done
python profiles/tools/lane_stamps.py 1024 2>&1 | tail -11
