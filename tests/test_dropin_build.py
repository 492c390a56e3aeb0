"""Drop-in check: the reference's own example drivers compile UNCHANGED against the host mirror
(csrc/include/cuddh.hpp) and link against libcuddh_amd.so.  Compile-only (hipcc cross-compiles without a
GPU); skipped where the reference tree is absent (it never travels to the GPU box)."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/examples")


@pytest.mark.skipif(not REF.exists(), reason="reference tree not present")
@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="no hipcc")
@pytest.mark.parametrize("name", ["DDH", "Poisson", "Helmholtz"])
def test_reference_example_compiles_unchanged(name, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = ROOT / "cuddhelmholtz_amd" / "lib"
    assert (lib / "libcuddh_amd.so").exists()
    cmd = [hipcc, "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-munsafe-fp-atomics",
           f"-I{ROOT / 'include'}", f"-I{ROOT / 'cuddhelmholtz_amd' / 'csrc' / 'include'}",
           f"-I{ROOT / 'cuddhelmholtz_amd' / 'csrc' / 'examples'}", f"-I{REF}",
           str(REF / f"{name}.cpp"), "-o", str(tmp_path / name), f"-L{lib}", "-lcuddh_amd"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    assert (tmp_path / name).exists()


@pytest.mark.skipif(not REF.exists(), reason="reference tree not present")
@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="no hipcc")
def test_reference_test_suite_compiles_unchanged(tmp_path):
    """The reference's own tests (tests/test.cpp and the files it drives), compiled where they lie.  The binary the GPU
    test runs (tests/test_gpu_native_drivers.py::test_reference_test_suite_runs_unchanged) is built by build_examples()."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = ROOT / "cuddhelmholtz_amd" / "lib"
    tests = REF.parent / "tests"
    srcs = sorted(tests.glob("*.cpp"))
    assert len(srcs) == 8
    cmd = [hipcc, "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-munsafe-fp-atomics",
           f"-I{ROOT / 'include'}", f"-I{ROOT / 'cuddhelmholtz_amd' / 'csrc' / 'include'}", f"-I{tests}",
           f'-DUNSTRUCTURED_SQUARE_MESH_DIR="{ROOT / "tests" / "golden" / "unstructured_square"}"',
           *map(str, srcs), "-o", str(tmp_path / "reference_tests"), f"-L{lib}", "-lcuddh_amd"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
