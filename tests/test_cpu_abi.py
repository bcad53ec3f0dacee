"""The C-ABI library loads and exports every symbol include/zopfli_amd.h declares (no compute calls)."""
import ctypes
import os
import re

from zopfli_amd._build import LIB, ROOT


def test_exports_match_header():
    assert os.path.exists(LIB), "libzopfli_amd.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(LIB)
    with open(os.path.join(ROOT, "include", "zopfli_amd.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = set(re.findall(r"\b((?:Zopfli|zmx_)\w+)\s*\(", text))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/zopfli_amd.h but not exported"


def test_options_layout():
    from zopfli_amd import ZopfliOptions, api
    lib = api.library()
    o = ZopfliOptions(1, 0, 1)
    lib.ZopfliInitOptions(ctypes.byref(o))
    assert (o.verbose, o.verbose_more, o.numiterations, o.blocksplitting, o.blocksplittinglast,
            o.blocksplittingmax) == (0, 0, 15, 1, 0, 15)
    assert ctypes.sizeof(ZopfliOptions) == 24
