"""CPU-side checks of the C-ABI: the library builds (hipcc cross-compiles without a GPU), loads,
and exports every symbol include/ofd.h declares; the ctypes table binds exactly that set."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def libpath():
    from opticalflowdiffusion_amd import build
    return build.build(verbose=False)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ofd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ofd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/ofd.h but not exported: {missing}"


def test_ctypes_table_matches_header(libpath):
    from opticalflowdiffusion_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    L = _lib.lib()
    assert L.ofd_version() >= 1
    assert L.ofd_last_error() is not None


def test_argument_errors_without_gpu(libpath):
    """argument validation happens before any HIP call: usable (and tested) on a CPU-only host"""
    from opticalflowdiffusion_amd import _lib
    L = _lib.lib()
    assert L.ofd_splat_workspace_bytes(2, 8, 8) == 16 + 2 * 256 * 4 + 8 * 2 * 8 * 8
    rc = L.ofd_splat_fwd(None, None, None, 1, 1, 8, 8, 3, 0, 0, 4, None, 0, None)      # null pointers
    assert rc == -1 and b"null" in L.ofd_last_error()
    rc = L.ofd_splat_fwd(None, None, None, 1, 1, 8, 8, 16, 0, 0, 4, None, 0, None)     # H // scale == 0
    assert rc == -1 and b"scale" in L.ofd_last_error()
    rc = L.ofd_warp_holes(None, None, 1, 1, 4, 4, 7, 1, None)
    assert rc == -1
    args = _lib.ConvArgs()
    assert L.ofd_conv_forward(ctypes.byref(args), None) == -1
    assert L.ofd_conv_gn_partial_count(2, 16, 64, 64) == 2 * 2 * 2 * 4 * 8 * 2
    assert L.ofd_conv_weight_elems(64, 16, 7) == 49 * 16 * 64


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from opticalflowdiffusion_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OfdError, match="no CPU or PyTorch fallback"):
        _lib.lib()
