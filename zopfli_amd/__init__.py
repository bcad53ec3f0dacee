"""zopfli_amd — MI355X-native drop-in for zopfli's LZ77 optimal-parse hot path.

The product is the C-ABI shared library ``libzopfli_amd.so`` (see
``include/zopfli_amd.h``); this package is the thin Python mirror of the
reference's interface used by the tests and by ``bench.py``:

    ZopfliOptions, ZopfliFormat, compress(), deflate()      # zopfli.h / deflate.h
    Context                                                 # the zmx_* device layer

There is no CPU fallback: importing works anywhere, but every call needs the
HIP library and a gfx950 device and raises otherwise.
"""
from .api import (FORMAT_DEFLATE, FORMAT_GZIP, FORMAT_ZLIB, Context, Dist, ZopfliOptions, compress, deflate,
                  deflate_part, library, last_timing)
from .datagen import generate

__all__ = ["ZopfliOptions", "FORMAT_GZIP", "FORMAT_ZLIB", "FORMAT_DEFLATE", "compress", "deflate",
           "deflate_part", "Context", "Dist", "library", "generate", "last_timing"]
