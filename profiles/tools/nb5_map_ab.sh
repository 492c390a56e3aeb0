for v in base nomap base nomap; do
  if [ $v = base ]; then L="X=1"; else L="CUDDH_AMD_LIBRARY_VARIANT=libcuddh_amd_$v.so"; fi
  echo "== $v (base = element -> local dof map re-read before the colour phases; nomap = held, spilled by the compiler)"
  env $L python3 profiles/tools/native_apply.py 768 5 20 2>&1 | grep ordering | tail -1 | cut -c1-230 | sed "s/^/768^2: /"
  env $L python3 profiles/tools/native_apply.py 0 5 20 5 2>&1 | grep ordering | tail -1 | cut -c1-230 | sed "s/^/irregular r=5: /"
done
