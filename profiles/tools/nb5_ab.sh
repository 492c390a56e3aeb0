for u in 0 1; do echo "== CUDDH_HELM_ULDS=$u  768^2"; CUDDH_HELM_ULDS=$u python3 profiles/tools/native_apply.py 768 5 20 2>&1 | grep -v "amdgpu.ids\|to_native"; done
for u in 0 1; do echo "== CUDDH_HELM_ULDS=$u  irregular r=5"; CUDDH_HELM_ULDS=$u python3 profiles/tools/native_apply.py 0 5 20 5 2>&1 | grep -v "amdgpu.ids\|to_native"; done
