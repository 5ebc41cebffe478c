"""CPU-side check that libcutie_hip.so loads and exports every symbol include/cutie_hip.h
declares, and that the ctypes table covers them (no compute calls without a GPU)."""
import ctypes
import os
import re

from cutie_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(REPO, "include", "cutie_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cutie_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 24
    lib = capi.lib()
    for s in syms:
        assert hasattr(lib, s), "libcutie_hip.so does not export %s" % s
        assert s in capi.SIGNATURES, "capi.SIGNATURES lacks %s" % s
    assert set(capi.SIGNATURES) == set(syms)
    assert lib.cutie_version() >= 100


def test_integration_table_names_every_symbol():
    """INTEGRATION.md's table (C entry point -> what it replaces in the reference) covers the whole ABI"""
    with open(os.path.join(REPO, "INTEGRATION.md")) as f:
        text = f.read()
    named = set(re.findall(r"`(cutie_[a-z0-9_]+)`", text))
    missing = [s for s in declared_symbols() if s not in named]
    assert not missing, "INTEGRATION.md does not mention %s" % missing


def test_argument_errors_are_reported_not_thrown():
    lib = capi.lib()
    rc = lib.cutie_grid_build(None, None, None, None, None, 1, 0, 4, 4, 7, None, None, None, None, None, None,
                              None, None, 0, None)
    assert rc == -1
    assert b"strategy" in lib.cutie_last_error()
    assert ctypes.sizeof(capi.ConvSeg) == 48
