"""Builds a second copy of the library with extra preprocessor defines on ONE kernel file, for same-box A/B runs:
  python profiles/tools/build_variant.py NAME helmholtz_fused.hip -DHELM_MFMA_GS=1 -DHELM_MFMA_GM=1
writes cuddhelmholtz_amd/lib/libcuddh_amd_NAME.so (git-ignored, shipped by gpurun); select it with
CUDDH_AMD_LIBRARY_VARIANT=libcuddh_amd_NAME.so.  All other objects are the default build's."""
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from cuddhelmholtz_amd import build as B  # noqa: E402

name, src_name, defines = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_native()
src = B.CSRC / "kernels" / src_name
obj = B.OBJ_DIR / f"{src_name}.{name}.o"
cmd = [B.hipcc(), *B.COMMON, *[f"-I{p}" for p in B.INCLUDES], *B.HIP_FLAGS, *B.EXTRA_FLAGS.get(src_name, []), *defines, "-c", str(src), "-o", str(obj)]
subprocess.run(cmd, check=True, capture_output=True)
hip, cpp = B._sources()
objs = [obj if s.name == src_name else B.OBJ_DIR / (s.name + ".o") for s in hip + cpp]
out = B.LIB_DIR / f"libcuddh_amd_{name}.so"
subprocess.run([B.hipcc(), "-shared", f"--offload-arch={B.ARCH}", "-o", str(out), *map(str, objs), "-ldl"], check=True, capture_output=True)
print(out)
