#!/bin/bash
# BLAS-1 / fused Gram-Schmidt stage kernels: requests of a tile issued together, the first tile before the coefficient is summed (base)
# vs the guarded form (oldblas), same box, alternating: small vectors (config 2: 1.18 M doubles), large ones, and GMRES(20) of config 2
python3 -m pytest tests/test_gpu_parity.py -q -k "blas1 or gmres or mgs" 2>&1 | tail -1
for v in base oldblas base oldblas; do
  if [ $v = base ]; then unset CUDDH_AMD_LIBRARY_VARIANT; else export CUDDH_AMD_LIBRARY_VARIANT=libcuddh_amd_$v.so; fi
  echo "######## $v"
  python3 profiles/tools/blas_rates.py 1182722 2>&1 | grep "n="
  python3 profiles/tools/blas_rates.py 18886658 2>&1 | grep "n="
  python3 profiles/tools/blas_rates.py 134217728 2>&1 | grep "n="
  python3 profiles/tools/gmres_helm.py 256 2>&1 | grep -i "matvec\|us per" | head -4
done
