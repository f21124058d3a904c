"""CPU checks of the C-ABI boundary: the HIP library builds for gfx950, loads, and exports
every symbol include/bmp.h declares; the ctypes table in bmp/_lib.py covers exactly those
symbols.  No compute call is made (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bmp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bmp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from bmp import _lib
    return _lib


def test_header_declares_entry_points():
    names = _declared()
    assert "bmp_msg_fwd" in names and "bmp_gru_bwd" in names and len(names) >= 15


def test_library_exports_every_declared_symbol(built):
    L = built.lib()
    for name in _declared():
        assert hasattr(L, name), f"{name} declared in include/bmp.h but not exported"


def test_ctypes_table_matches_header(built):
    assert sorted(built.SIGNATURES) == _declared()


def test_argument_counts_match_header(built):
    src = open(os.path.join(ROOT, "include", "bmp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, (_, args) in built.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, src)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} params, ctypes table {len(args)}"


def test_library_is_gfx950_only(built):
    blob = open(built.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_"):
        assert other not in blob


def test_missing_library_fails_loudly(monkeypatch, built):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libbmp_hip.so")
    with pytest.raises(built.BmpLibraryError):
        built.lib()
