"""Builds the native library (HIP kernels + C ABI + host C++ mirror) for gfx950.

`python -m cuddhelmholtz_amd.build` or `build_native()`; hipcc cross-compiles
without a GPU.  Objects land in build/, the shared library in
cuddhelmholtz_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "cuddhelmholtz_amd" / "csrc"
LIB_DIR = ROOT / "cuddhelmholtz_amd" / "lib"
LIB_PATH = LIB_DIR / "libcuddh_amd.so"
OBJ_DIR = ROOT / "build" / "obj"
ARCH = "gfx950"

INCLUDES = [ROOT / "include", CSRC / "include", CSRC / "kernels"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-Wall", "-Wno-unused-parameter", "-Wno-unused-function"]
HIP_FLAGS = ["-x", "hip", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-ffp-contract=fast"]

# per-file extra flags: SLP-packing the fp32 time-stepping loop into v_pk_* is an anti-lever on gfx950
# (a v_pk_fma_f32 is no faster than two v_fma_f32 and blocks DPP folding)
# -amdgpu-mfma-vgpr-form: keep the 4x4x1 MFMA accumulators in VGPRs (gfx950 has a unified file) instead of
# shuttling them through AGPRs with v_accvgpr_read/write
EXTRA_FLAGS = {"ddh.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]}

# host sources that contain device code (device lambdas) and must be compiled as HIP
HIP_HOST_SOURCES = {"capi.cpp"}


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found")
    return exe


def _sources():
    hip = sorted((CSRC / "kernels").glob("*.hip"))
    cpp = sorted((CSRC / "src").glob("*.cpp"))
    return hip, cpp


def _needs_rebuild(obj: Path, src: Path, headers_mtime: float) -> bool:
    if not obj.exists():
        return True
    m = obj.stat().st_mtime
    return m < src.stat().st_mtime or m < headers_mtime


# what the last build()/build_native()/build_examples() calls of this process did: written to build/BUILD_INFO.json so that a
# reader of the tree can see whether objects were compiled or reused
BUILD_LOG: dict = {"compiled": [], "reused": [], "linked": [], "drivers_built": [], "drivers_reused": []}


def write_build_info() -> Path:
    import json
    import time

    def run(*cmd):
        try:
            return subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT).stdout.strip()
        except OSError:
            return ""

    hipcc_version = next((ln for ln in run(hipcc(), "--version").splitlines() if "HIP version" in ln or "clang version" in ln), "")
    info = {
        "written_at": time.strftime("%Y-%m-%dT%H:%M:%S%z"),
        "arch": ARCH,
        "hipcc": hipcc_version,
        "git_head": run("git", "rev-parse", "HEAD"),
        "git_dirty": bool(run("git", "status", "--porcelain", "--", "cuddhelmholtz_amd", "include", "oracle")),
        "build_mode": "compiled" if BUILD_LOG["compiled"] or BUILD_LOG["drivers_built"] else "reused (every object newer than its sources and headers)",
        **{k: sorted(set(v)) for k, v in BUILD_LOG.items()},
        "library": str(LIB_PATH.relative_to(ROOT)),
        "library_bytes": LIB_PATH.stat().st_size if LIB_PATH.exists() else 0,
        "reference_tree_present": REFERENCE_EXAMPLES.exists(),
    }
    out = ROOT / "build" / "BUILD_INFO.json"
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps(info, indent=1) + "\n")
    return out


def _compile(src: Path, as_hip: bool, headers_mtime: float, verbose: bool) -> Path:
    obj = OBJ_DIR / (src.name + ".o")
    if not _needs_rebuild(obj, src, headers_mtime):
        BUILD_LOG["reused"].append(src.name)
        return obj
    BUILD_LOG["compiled"].append(src.name)
    cmd = [hipcc(), *COMMON, *[f"-I{p}" for p in INCLUDES]]
    if as_hip:
        cmd += HIP_FLAGS
    cmd += EXTRA_FLAGS.get(src.name, [])
    cmd += ["-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"compilation of {src.name} failed:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build_native(verbose: bool = False, jobs: int = 6) -> Path:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    headers = [p for d in INCLUDES for p in d.rglob("*.h*")]
    headers_mtime = max(p.stat().st_mtime for p in headers)
    hip, cpp = _sources()
    work = [(s, True) for s in hip] + [(s, s.name in HIP_HOST_SOURCES) for s in cpp]
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda sw: _compile(sw[0], sw[1], headers_mtime, verbose), work))
    newest = max(o.stat().st_mtime for o in objs)
    if not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest:
        # (RCCL is bound at run time by csrc/src/multigpu.cpp, not linked: a host process may hold its own copy)
        cmd = [hipcc(), "-shared", f"--offload-arch={ARCH}", "-o", str(LIB_PATH), *map(str, objs), "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        BUILD_LOG["linked"].append(LIB_PATH.name)
    write_build_info()
    return LIB_PATH


EXAMPLES_DIR = ROOT / "build" / "examples"
REFERENCE_EXAMPLES = Path("/root/reference/examples")


def build_examples(verbose: bool = False) -> list[Path]:
    """Compiles the native drivers against csrc/include/cuddh.hpp: this repository's ddh_solve.cpp and, when the
    reference tree is present, the reference's own examples/{DDH,Poisson,Helmholtz}.cpp UNCHANGED (drop-in check).
    Binaries go to build/examples/ (git-ignored, shipped to the GPU box)."""
    EXAMPLES_DIR.mkdir(parents=True, exist_ok=True)
    jobs = [(CSRC / "examples" / "ddh_solve.cpp", "ddh_solve"), (CSRC / "examples" / "helmholtz_solve.cpp", "helmholtz_solve")]
    if REFERENCE_EXAMPLES.exists():
        jobs += [(REFERENCE_EXAMPLES / f"{n}.cpp", f"{n}_reference_driver") for n in ("DDH", "Poisson", "Helmholtz")]
    # the reference's OWN test suite (tests/test.cpp + quadrature_rule, basis, gmres, linalg, mass, stiffness and the text-mesh
    # loader), every file compiled where it lies and unchanged, against csrc/include/cuddh.hpp; the mesh fixture it reads is
    # the copy under tests/golden/ (data)
    REFERENCE_TESTS = REFERENCE_EXAMPLES.parent / "tests"
    if REFERENCE_TESTS.exists():
        jobs.append((sorted(REFERENCE_TESTS.glob("*.cpp")), "reference_tests_driver"))
    outs = []
    for src, name in jobs:
        out = EXAMPLES_DIR / name
        if isinstance(src, list):
            env_hdr = CSRC / "examples" / "reference_tests_env.hpp"
            if out.exists() and out.stat().st_mtime > max(max(f.stat().st_mtime for f in src), LIB_PATH.stat().st_mtime, env_hdr.stat().st_mtime):
                outs.append(out)
                BUILD_LOG["drivers_reused"].append(name)
                continue
            BUILD_LOG["drivers_built"].append(name)
            cmd = [hipcc(), "-O2", "-std=c++17", "-x", "hip", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
                   *[f"-I{p}" for p in INCLUDES], f"-I{src[0].parent}",
                   # the reference reads `std::string dir = UNSTRUCTURED_SQUARE_MESH_DIR;` (tests/load_unstructured_square.cpp:13): the
                   # macro is an expression here, resolved at run time from the binary's own location (or CUDDH_MESH_DIR), so the
                   # driver works from any checkout path
                   "-include", str(CSRC / "examples" / "reference_tests_env.hpp"), "-DUNSTRUCTURED_SQUARE_MESH_DIR=cuddh_reference_mesh_dir()",
                   *map(str, src), "-o", str(out),
                   f"-L{LIB_DIR}", "-lcuddh_amd", "-Wl,-rpath,$ORIGIN/../../cuddhelmholtz_amd/lib"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"building {name} failed:\n{r.stderr}")
            outs.append(out)
            continue
        if out.exists() and out.stat().st_mtime > max(src.stat().st_mtime, LIB_PATH.stat().st_mtime):
            outs.append(out)
            BUILD_LOG["drivers_reused"].append(name)
            continue
        BUILD_LOG["drivers_built"].append(name)
        cmd = [hipcc(), "-O2", "-std=c++17", "-x", "hip", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
               *[f"-I{p}" for p in INCLUDES], f"-I{CSRC / 'examples'}", f"-I{REFERENCE_EXAMPLES}", str(src), "-o", str(out),
               f"-L{LIB_DIR}", "-lcuddh_amd", "-Wl,-rpath,$ORIGIN/../../cuddhelmholtz_amd/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"building {name} failed:\n{r.stderr}")
        outs.append(out)
    write_build_info()
    return outs


if __name__ == "__main__":
    print(build_native(verbose="-v" in sys.argv))
    print(*build_examples(verbose="-v" in sys.argv))
