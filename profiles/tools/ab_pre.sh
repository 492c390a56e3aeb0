python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -k "patch_sizes or fused or golden or tiny or plan" 2>&1 | tail -2
for i in 1 2; do CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 1024 4 2>&1 | grep "fused"; done
CUDDH_HELM_PRE=1 CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 1024 4 2>&1 | grep "fused"
CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 1024 3 2>&1 | grep "fused"
CUDDH_PLAN_AFFINE=0 python profiles/tools/time_operators.py 1024 2 2>&1 | grep "fused"
python profiles/tools/unstructured_apply.py 6 4 2>&1 | grep fused
python profiles/tools/lane_stamps.py 1024 2>&1 | tail -10
