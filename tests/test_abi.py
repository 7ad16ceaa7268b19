"""CPU tests of the drop-in boundary: libgmrm_hip.so loads, exports every symbol
include/gmrm_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

import gmrm_amd
from gmrm_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "gmrm_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gmrm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = gmrm_amd.load_library()
    names = _declared()
    assert len(names) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", str(gmrm_amd.library_path())], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (gmrm_[a-z0-9_]+)", out))
    for n in names:
        assert n in exported, f"{n} declared in gmrm_hip.h but not exported"
        assert hasattr(lib, n)
    assert set(_lib.SIGNATURES) == set(names), "ctypes table and header disagree"


def test_abi_version_and_error_string():
    lib = gmrm_amd.load_library()
    assert lib.gmrm_abi_version() == 1
    h = C.c_void_p()
    rc = lib.gmrm_ctx_create(C.byref(h), 0, 1, 10, 10, 0, 1)      # N = 1 is invalid
    assert rc == -1
    assert b"dimension" in lib.gmrm_last_error()


def test_no_cpu_fallback_without_device():
    lib = gmrm_amd.load_library()
    if lib.gmrm_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(gmrm_amd.GmrmError) as ei:
        gmrm_amd.Context(1000, 10)
    assert ei.value.code == -2          # GMRM_ENODEV


def test_product_does_not_touch_the_oracle():
    """The shipped path must not import, link or call anything under oracle/."""
    bad = re.compile(r"import\s+oracle|from\s+oracle|liborc|orc_[a-z]|oracle/")
    for p in (ROOT / "gmrm_amd").rglob("*"):
        if p.suffix in (".py", ".cpp", ".hip", ".h") and "_build" not in p.parts:
            assert not bad.search(p.read_text()), f"{p} reaches into oracle/"
    out = subprocess.run(["ldd", str(gmrm_amd.library_path())], capture_output=True, text=True).stdout
    assert "liborc" not in out


def test_block_of_markers_matches_reference_rule():
    # bayes.cpp:903-925
    from gmrm_amd import block_of_markers
    assert [block_of_markers(10, 3, r) for r in range(3)] == [(0, 4, 4), (4, 3, 4), (7, 3, 4)]
    assert [block_of_markers(8, 2, r) for r in range(2)] == [(0, 4, 4), (4, 4, 4)]
