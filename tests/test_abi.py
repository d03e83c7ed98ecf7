"""The C-ABI library loads on a machine without a GPU and exports every symbol include/pplhip.h declares."""
import ctypes
import os
import re

from tests.conftest import ROOT, load_pplhip


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pplhip.h")).read()
    return sorted(set(re.findall(r"PPLHIP_API\s+[\w\s\*]+?\b(pplhip_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    m = load_pplhip()
    assert header_symbols() == sorted(m.SYMBOLS)


def test_library_exports_every_symbol():
    m = load_pplhip()
    assert os.path.exists(m.LIB_PATH), "libpplhip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(m.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), s
    assert lib.pplhip_version() >> 16 == 1


def test_struct_layouts_match_oracle():
    """pplhip_model_desc / pplhip_step and the oracle's ref_model_desc / ref_step are the same bytes."""
    from oracle import ref
    m = load_pplhip()
    assert ctypes.sizeof(m.ModelDesc) == ctypes.sizeof(ref.ModelDesc) == 68
    assert ctypes.sizeof(m.Step) == ctypes.sizeof(ref.Step)
    assert [f[0] for f in m.ModelDesc._fields_] == [f[0] for f in ref.ModelDesc._fields_]
    assert [f[0] for f in m.Step._fields_] == [f[0] for f in ref.Step._fields_]


def test_rope_table_matches_oracle():
    """host-only entry point: the cos/sin table the device uses equals the oracle's bit for bit."""
    import numpy as np
    from oracle import ref
    m = load_pplhip()
    a = np.empty((257, 64), dtype=np.float32)
    b = np.empty((257, 64), dtype=np.float32)
    assert m.lib().pplhip_build_rope_table(a.ctypes.data, 257, 64, 10000.0) == 0
    ref.lib().ref_build_rope_table(b.ctypes.data, 257, 64, 10000.0)
    assert (a == b).all()
