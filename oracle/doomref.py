"""ctypes binding for the ORACLE (oracle/libdoomref.so).  TEST INFRASTRUCTURE ONLY: importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from the product package."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libdoomref.so")


class View(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in "x y angle floor_height cos_a sin_a cos_na sin_na timestamp".split()] + \
               [("trig_valid", ctypes.c_int32)]


class ListColumn(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in "x clipped_top_y clipped_bottom_y bottom_y top_y".split()]


class ListRender(ctypes.Structure):
    _fields_ = [("texture", ctypes.c_char_p)] + [(n, ctypes.c_int16) for n in "light_level offset_x offset_y reserved".split()] + \
               [(n, ctypes.c_float) for n in "line_start_x line_start_y line_end_x line_end_y start_offset".split()] + \
               [("start_x", ctypes.c_int32), ("end_x", ctypes.c_int32), ("bottom_height", ctypes.c_float), ("top_height", ctypes.c_float),
                ("first_column", ctypes.c_uint32), ("n_columns", ctypes.c_uint32)]


class ListVisplane(ctypes.Structure):
    _fields_ = [("flat", ctypes.c_char_p)] + [(n, ctypes.c_int16) for n in "height light_level left right".split()] + [("first_entry", ctypes.c_uint32)]


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in "n_records n_columns n_visplanes n_mobj_records".split()] + \
               [(n, ctypes.c_int64) for n in "wall_pixels flat_pixels sky_pixels masked_pixels mobj_pixels".split()]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("doomref.c", "doomref.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.dr_load.restype = ctypes.c_void_p
        L.dr_load.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.dr_free.argtypes = [ctypes.c_void_p]
        L.dr_last_error.restype = ctypes.c_char_p
        L.dr_render.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(View), ctypes.c_void_p, ctypes.c_int]
        L.dr_player_start.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_float)] * 3
        L.dr_floor_height_at.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_float)]
        L.dr_diminish_color.argtypes = [ctypes.c_char_p, ctypes.c_int16, ctypes.c_int16, ctypes.c_char_p]
        L.dr_f32_as_i16.restype = ctypes.c_int16
        L.dr_f32_as_i16.argtypes = [ctypes.c_float]
        L.dr_f32_as_i32.restype = ctypes.c_int32
        L.dr_f32_as_i32.argtypes = [ctypes.c_float]
        L.dr_f32_as_u8.restype = ctypes.c_uint8
        L.dr_f32_as_u8.argtypes = [ctypes.c_float]
        L.dr_constants.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.dr_last_stats.argtypes = [ctypes.POINTER(Stats)]
        L.dr_draw_lists.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(View), ctypes.POINTER(ListRender), ctypes.c_int,
                                    ctypes.POINTER(ListColumn), ctypes.POINTER(ListVisplane), ctypes.c_int, ctypes.POINTER(ctypes.c_int16),
                                    ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_void_p]
        L.dr_sector_count.argtypes = [ctypes.c_void_p]
        L.dr_set_sector_light.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int16]
        L.dr_mobj_count.argtypes = [ctypes.c_void_p]
        L.dr_set_mobj_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint8, ctypes.c_int]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


class Scene:
    """dr_scene handle: the reference's Game::new minus SDL (src/game.rs:142-167)."""

    def __init__(self, wad: bytes, map_name: str = "e1m1"):
        self._h = lib().dr_load(wad, len(wad), map_name.encode())
        if not self._h:
            raise OracleError(lib().dr_last_error().decode())

    def player_start(self):
        x, y, a = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        if lib().dr_player_start(self._h, x, y, a):
            raise OracleError(lib().dr_last_error().decode())
        return x.value, y.value, a.value

    def floor_height_at(self, x: float, y: float, default: float = 0.0) -> float:
        h = ctypes.c_float(default)
        lib().dr_floor_height_at(self._h, x, y, h)
        return h.value

    def render(self, W: int, H: int, view, flags: int = 0, out=None):
        """view: camera-path record of 8 (or 9) f32: x, y, angle, cos, sin, cos(-a), sin(-a), floor_height[, timestamp]
        (camera_path.view_record) -> bytes (3*W*H)."""
        x, y, a, c, s, cn, sn, fh = [float(t) for t in view[:8]]
        v = View(x, y, a, fh, c, s, cn, sn, float(view[8]) if len(view) > 8 else 0.0, 1)
        buf = out if out is not None else ctypes.create_string_buffer(3 * W * H)
        ptr = ctypes.cast(buf, ctypes.c_void_p) if out is None else ctypes.c_void_p(out)
        if lib().dr_render(self._h, W, H, ctypes.byref(v), ptr, flags):
            raise OracleError(lib().dr_last_error().decode())
        return buf.raw if out is None else None

    def draw_lists(self, W: int, H: int, view, lists: dict) -> bytes:
        """Replay hand-built records (tests/np_mappers.py list dict) through the oracle's pixel functions -> bytes (3*W*H)."""
        x, y, a, c, s, cn, sn, fh = [float(t) for t in view[:8]]
        v = View(x, y, a, fh, c, s, cn, sn, 0.0, 1)
        cols = (ListColumn * max(1, len(lists["columns"])))(*[ListColumn(*[int(t) for t in col]) for col in lists["columns"]])
        rs = (ListRender * max(1, len(lists["renders"])))()
        for i, r in enumerate(lists["renders"]):
            rs[i] = ListRender(r["texture"].encode(), r["light_level"], r["offset_x"], r["offset_y"], 0, *[float(t) for t in r["line"]], float(r["start_offset"]),
                               r["start_x"], r["end_x"], float(r["bottom_height"]), float(r["top_height"]), r["first_column"], r["n_columns"])
        vs = (ListVisplane * max(1, len(lists["visplanes"])))()
        tb = []
        for i, p in enumerate(lists["visplanes"]):
            vs[i] = ListVisplane(p["flat"].encode(), p["height"], p["light_level"], p["left"], p["right"], len(tb) // 2)
            for (t, b) in p["tb"]:
                tb += [t, b]
        tba = (ctypes.c_int16 * max(1, len(tb)))(*tb)
        order = (ctypes.c_uint32 * max(1, 2 * len(lists["order"])))(*[t for pair in lists["order"] for t in pair])
        buf = ctypes.create_string_buffer(3 * W * H)
        if lib().dr_draw_lists(self._h, W, H, ctypes.byref(v), rs, len(lists["renders"]), cols, vs, len(lists["visplanes"]), tba, order, len(lists["order"]),
                               ctypes.cast(buf, ctypes.c_void_p)):
            raise OracleError(lib().dr_last_error().decode())
        return buf.raw

    def sector_count(self) -> int:
        return lib().dr_sector_count(self._h)

    def set_sector_light(self, sector: int, light: int):
        if lib().dr_set_sector_light(self._h, sector, light):
            raise OracleError(lib().dr_last_error().decode())

    def mobj_count(self) -> int:
        return lib().dr_mobj_count(self._h)

    def set_mobj_state(self, mobj: int, sprite, frame: int = 0, full_bright: bool = False):
        if lib().dr_set_mobj_state(self._h, mobj, sprite.encode() if sprite else None, frame, int(full_bright)):
            raise OracleError(lib().dr_last_error().decode())

    def stats(self) -> dict:
        st = Stats()
        lib().dr_last_stats(ctypes.byref(st))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def close(self):
        if self._h:
            lib().dr_free(self._h)
            self._h = None
