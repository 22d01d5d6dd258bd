"""ctypes binding of the CPU test harness tests/emul/libdgemul.so (product host logic + kernel bodies on the CPU)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "emul", "libdgemul.so")
_lib = None


class DgView(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in "x y angle floor_height cos_a sin_a cos_na sin_na timestamp".split()] + \
               [("trig_valid", ctypes.c_int32)]


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "emul"), "-s"])
        L = ctypes.CDLL(_LIB)
        L.emul_load.restype = ctypes.c_void_p
        L.emul_load.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.emul_free.argtypes = [ctypes.c_void_p]
        L.emul_last_error.restype = ctypes.c_char_p
        L.emul_render.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(DgView), ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.emul_render_fe.argtypes = L.emul_render.argtypes
        L.emul_render_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(DgView), ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                        ctypes.c_uint32, ctypes.c_void_p]
        L.emul_fs_frame.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(DgView), ctypes.POINTER(ctypes.c_uint64)]
        L.emul_fs_no_cl_rows.argtypes = [ctypes.c_int]
        L.emul_fs_no_cl_rows.restype = None
        L.emul_sprite_frame.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint8]
        L.emul_set_sector_light.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int16]
        L.emul_set_mobj_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint8, ctypes.c_int]
        _lib = L
    return _lib


class EmulScene:
    def __init__(self, wad: bytes, map_name="e1m1"):
        self._h = lib().emul_load(wad, len(wad), map_name.encode())
        if not self._h:
            raise RuntimeError(lib().emul_last_error().decode())

    def set_sector_light(self, sector, light):
        lib().emul_set_sector_light(self._h, sector, light)

    def set_mobj_state(self, mobj, sprite, frame=0, full_bright=False):
        if lib().emul_set_mobj_state(self._h, mobj, sprite.encode() if sprite else None, frame, int(full_bright)):
            raise RuntimeError(lib().emul_last_error().decode())

    def render(self, W, H, rec, timestamp=0.0):
        v = DgView(float(rec[0]), float(rec[1]), float(rec[2]), float(rec[7]), float(rec[3]), float(rec[4]), float(rec[5]), float(rec[6]),
                   float(timestamp), 1)
        buf = np.empty(3 * W * H, dtype=np.uint8)
        st = (ctypes.c_uint64 * 4)()
        rc = lib().emul_render(self._h, W, H, ctypes.byref(v), buf.ctypes.data_as(ctypes.c_void_p), st)
        if rc:
            raise RuntimeError(f"emul rc {rc}: {lib().emul_last_error().decode()}")
        return buf.tobytes(), list(st)

    def sprite_frame(self, sprite, frame=0):
        sf = lib().emul_sprite_frame(self._h, sprite.encode(), frame)
        if sf < 0:
            raise RuntimeError(lib().emul_last_error().decode())
        return sf

    def render_state(self, W, H, rec, lights, mobjs, timestamp=0.0):
        """One view with a game-state snapshot: lights = [(sector, level)], mobjs = [(mobj, sprite_frame or -1, full_bright)]."""
        v = DgView(float(rec[0]), float(rec[1]), float(rec[2]), float(rec[7]), float(rec[3]), float(rec[4]), float(rec[5]), float(rec[6]),
                   float(timestamp), 1)
        la = np.array([[s, l] for s, l in lights], dtype=np.int32).reshape(-1, 2)
        ma = np.array([[m, sf, fb, 0] for m, sf, fb in mobjs], dtype=np.int32).reshape(-1, 4)
        buf = np.empty(3 * W * H, dtype=np.uint8)
        rc = lib().emul_render_state(self._h, W, H, ctypes.byref(v), la.ctypes.data, len(la), ma.ctypes.data, len(ma), buf.ctypes.data)
        if rc:
            raise RuntimeError(f"emul rc {rc}: {lib().emul_last_error().decode()}")
        return buf.tobytes()

    def fs_frame(self, W, H, rec, timestamp=0.0):
        """The device seg walk's bodies (fs_frame.h: dg_fs_segs / dg_fs_frame) on the CPU for one view, compared inside the
        harness with the host walker's parts mode record by record.  -> (rc, stats): rc 0 = identical records, 1 = the host walker
        refuses the frame and the device walk flagged it, 2 = the device walk gave the frame up because it exceeds a capacity (the host
        redoes it; stats[4] says which: 1 parts, 2 candidates, 4 sprites, 8 sky parts, 16 part bins, 32 sprite bins), 3 = it gave the frame
        up for no such reason; stats = [parts, sprites, sky slots, flags, capacities exceeded, candidates].  Raises on any mismatch."""
        v = DgView(float(rec[0]), float(rec[1]), float(rec[2]), float(rec[7]), float(rec[3]), float(rec[4]), float(rec[5]), float(rec[6]),
                   float(timestamp), 1)
        st = (ctypes.c_uint64 * 6)()
        rc = lib().emul_fs_frame(self._h, W, H, ctypes.byref(v), st)
        if rc < 0:
            raise RuntimeError(f"emul_fs_frame rc {rc}: {lib().emul_last_error().decode()}")
        return rc, list(st)

    def render_fe(self, W, H, rec, timestamp=0.0):
        """Same frame through the device column walk's bodies (fe_core.h) on the CPU.  stats = [spans, parts, sprites, overflow
        flags, span-list-identical-to-host-path, sky gap entries]."""
        v = DgView(float(rec[0]), float(rec[1]), float(rec[2]), float(rec[7]), float(rec[3]), float(rec[4]), float(rec[5]), float(rec[6]),
                   float(timestamp), 1)
        buf = np.empty(3 * W * H, dtype=np.uint8)
        st = (ctypes.c_uint64 * 6)()
        rc = lib().emul_render_fe(self._h, W, H, ctypes.byref(v), buf.ctypes.data_as(ctypes.c_void_p), st)
        if rc:
            raise RuntimeError(f"emul rc {rc}: {lib().emul_last_error().decode()}")
        return buf.tobytes(), list(st)
