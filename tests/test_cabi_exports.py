"""The C-ABI shared library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import subprocess

import cuddhelmholtz_amd._native as N


def test_library_loads_and_reports_no_gpu_gracefully():
    assert N.LIB_PATH.exists()
    assert N.lib.cuddh_hip_device_count() >= 0


def test_every_declared_symbol_is_exported():
    declared = N.declared_symbols()
    assert len(declared) > 100
    out = subprocess.run(["nm", "-D", "--defined-only", str(N.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    for s in declared:
        assert isinstance(getattr(N.lib, s), ctypes._CFuncPtr)


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    import pathlib
    import re

    pat = re.compile(r"import\s+oracle|from\s+oracle|liboracle|oracle/|oracle\.|#include[^\n]*oracle|orc_[a-z]")
    root = pathlib.Path(N.ROOT) / "cuddhelmholtz_amd"
    for p in list(root.rglob("*")) + [pathlib.Path(N.ROOT) / "include" / "cuddh_hip.h", pathlib.Path(N.ROOT) / "include" / "cuddh_capi.h"]:
        if p.suffix in {".py", ".cpp", ".hpp", ".hip", ".h", ".inc"}:
            assert not pat.search(p.read_text()), p
