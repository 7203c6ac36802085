"""The C-ABI shared library loads and exports every symbol include/bdetr.h declares
(no compute calls: this runs without a GPU)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "bdetr.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bdetr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from boosted_detr_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 40
    assert sorted(_lib.SIGNATURES) == syms, set(syms) ^ set(_lib.SIGNATURES)
    h = _lib.lib()           # binds every symbol or raises
    assert h.bdetr_abi_version() == 8
    for s in syms:
        assert hasattr(h, s)


def test_no_oracle_import_in_product():
    """The product package never imports the oracle (it is test infrastructure)."""
    for p in (ROOT / "boosted_detr_amd").rglob("*.py"):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), p
