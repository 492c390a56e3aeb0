#!/bin/bash
# same-box A/B of DDH kernel variants: ddh_ab.sh "nx,nb,kernel" VARIANT_OR_ENV...   (a token with '=' is an env assignment for one run,
# anything else names cuddhelmholtz_amd/lib/libcuddh_amd_<token>.so built by profiles/tools/build_variant.py; "base" = the default library)
SPEC=$1; shift
for v in "$@"; do
  if [[ "$v" == *=* ]]; then
    echo "== $v"; env "$v" python3 profiles/tools/ddh_rates.py $SPEC $SPEC 2>&1 | grep "ms per action"
  elif [[ "$v" == base ]]; then
    echo "== base"; python3 profiles/tools/ddh_rates.py $SPEC $SPEC 2>&1 | grep "ms per action"
  else
    echo "== variant $v"; CUDDH_AMD_LIBRARY_VARIANT=libcuddh_amd_$v.so python3 profiles/tools/ddh_rates.py $SPEC $SPEC 2>&1 | grep "ms per action"
  fi
done
