"""The C-ABI shared library builds, loads (no GPU needed) and exports every symbol include/vist3a_hip.h declares."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    txt = (ROOT / "include" / "vist3a_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(v3a_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_all_declared_symbols():
    from vist3a_amd import build, lib
    path = build.build(verbose=False)
    assert path.exists()
    dll = ctypes.CDLL(str(path))
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/vist3a_hip.h but not exported"
    # the Python binding table covers the header exactly
    assert sorted(lib.SYMBOLS) == names
    L = lib.load()
    assert L.v3a_abi_version() >= 1
    assert b"gfx950" in L.v3a_build_info()


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before any launch (safe on a GPU-less host)."""
    from vist3a_amd import lib
    L = lib.load()
    assert L.v3a_gemm_bf16_nt(None, None) == -1
    a = lib.GemmArgs()
    a.A = a.B = a.C = 1
    a.M, a.N, a.K = 8, 8, 7  # K % 64 != 0
    a.lda = a.ldb = a.ldc = 8
    assert L.v3a_gemm_bf16_nt(ctypes.byref(a), None) == -2
    at = lib.AttnArgs()
    at.q = at.k = at.vt = at.o = 1
    at.B = at.H = at.Nq = at.Nk = 1
    at.D = 96
    assert L.v3a_attention_fwd_bf16(ctypes.byref(at), None) == -2
    assert L.v3a_layernorm(None, None) == -1
    assert L.v3a_rmsnorm_rope(None, None) == -1


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from vist3a_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    import pytest
    with pytest.raises(lib.HipLibraryError):
        lib.load(tmp_path / "nope.so")
    monkeypatch.setattr(lib, "_lib", None)
