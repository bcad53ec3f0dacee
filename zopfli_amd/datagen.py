"""Seeded synthetic corpora (classes T X R Z B P M), see csrc/tools/datagen.c."""
import ctypes
import os

from ._build import DATAGEN, build_datagen

_lib = None
DEFAULT_SEED = {"T": 1, "X": 2, "R": 3, "Z": 4, "B": 5, "P": 6, "M": 7}


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(DATAGEN):
            build_datagen()
        _lib = ctypes.CDLL(DATAGEN)
        _lib.zopfli_amd_datagen.argtypes = [ctypes.c_char, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t]
        _lib.zopfli_amd_datagen.restype = ctypes.c_int
    return _lib


def generate(cls, n, seed=None):
    """n bytes of synthetic class `cls` (one of T X R Z B P M)."""
    lib = _load()
    buf = (ctypes.c_ubyte * n)()
    if lib.zopfli_amd_datagen(cls.encode(), DEFAULT_SEED[cls] if seed is None else seed, buf, n) != 0:
        raise RuntimeError("datagen failed")
    return bytes(buf)
