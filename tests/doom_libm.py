import ctypes
import ctypes.util

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.cosf.restype = ctypes.c_float
_libm.cosf.argtypes = [ctypes.c_float]
_libm.sinf.restype = ctypes.c_float
_libm.sinf.argtypes = [ctypes.c_float]


def same_libm_as_fixture(path) -> bool:
    for r in path[::97]:
        if np.float32(_libm.cosf(r[2])) != r[3] or np.float32(_libm.sinf(r[2])) != r[4]:
            return False
    return True
