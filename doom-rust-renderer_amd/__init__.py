"""doom-rust-renderer_amd — MI355X-native column/span rasteriser behind the reference's Pixels/Renderer draw API.

This Python module is only the test/bench harness binding (ctypes) of the C-ABI in ``include/doomgpu.h``;
the product is ``libdoomgpu.so`` (hand-written HIP kernels + C++ host list generation, ``csrc/``).
Import with ``importlib.import_module("doom-rust-renderer_amd")`` (the directory name carries a hyphen).

There is no CPU fallback anywhere in this package: without the built library every call raises, and without
a gfx950 device ``Context`` raises ``DoomGpuError`` (DG_ERR_NO_DEVICE).  The oracle under ``oracle/`` is never
imported from here.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DOOMGPU_LIB") or os.path.join(_HERE, "libdoomgpu.so")   # override only for kernel-variant experiments
INCLUDE = os.path.join(os.path.dirname(_HERE), "include", "doomgpu.h")

DG_OK, DG_ERR_INVALID, DG_ERR_NO_DEVICE, DG_ERR_HIP, DG_ERR_WAD, DG_ERR_RENDER, DG_ERR_CAPACITY = 0, -1, -2, -3, -4, -5, -6
DG_FE_AUTO, DG_FE_HOST, DG_FE_DEVICE, DG_FE_DEVICE_SEGS = 0, 1, 2, 3


class DoomGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"doomgpu error {code}: {msg}")
        self.code = code


class DgView(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in "x y angle floor_height cos_a sin_a cos_na sin_na timestamp".split()] + \
               [("trig_valid", ctypes.c_int32)]


class DgSectorLight(ctypes.Structure):
    _fields_ = [("sector", ctypes.c_int32), ("light_level", ctypes.c_int32)]


class DgMobjState(ctypes.Structure):
    _fields_ = [("mobj", ctypes.c_int32), ("sprite_frame", ctypes.c_int32), ("full_bright", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class DgViewState(ctypes.Structure):
    _fields_ = [("lights", ctypes.POINTER(DgSectorLight)), ("n_lights", ctypes.c_uint32),
                ("mobjs", ctypes.POINTER(DgMobjState)), ("n_mobjs", ctypes.c_uint32)]


class DgConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in "device width height max_batch slots host_threads front_end".split()]


class DgTiming(ctypes.Structure):
    _fields_ = [("setup_ms", ctypes.c_float), ("raster_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("host_ms", ctypes.c_float),
                ("n_spans", ctypes.c_uint64), ("n_frames", ctypes.c_uint64), ("covered_pixels", ctypes.c_uint64),
                ("n_walls", ctypes.c_uint64), ("n_planes", ctypes.c_uint64), ("list_bytes", ctypes.c_uint64),
                ("front_end", ctypes.c_int32)]


class DgBitmapColumn(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int16) for n in "x clipped_top_y clipped_bottom_y bottom_y top_y".split()]


class DgBitmapRender(ctypes.Structure):
    _fields_ = [("bitmap", ctypes.c_int32), ("light_level", ctypes.c_int16), ("offset_x", ctypes.c_int16),
                ("offset_y", ctypes.c_int16), ("reserved", ctypes.c_int16),
                ("line_start_x", ctypes.c_float), ("line_start_y", ctypes.c_float), ("line_end_x", ctypes.c_float),
                ("line_end_y", ctypes.c_float), ("start_offset", ctypes.c_float), ("start_x", ctypes.c_int32),
                ("end_x", ctypes.c_int32), ("bottom_height", ctypes.c_float), ("top_height", ctypes.c_float),
                ("first_column", ctypes.c_uint32), ("n_columns", ctypes.c_uint32)]


class DgVisplane(ctypes.Structure):
    _fields_ = [("flat", ctypes.c_int32), ("height", ctypes.c_int16), ("light_level", ctypes.c_int16),
                ("left", ctypes.c_int16), ("right", ctypes.c_int16), ("first_entry", ctypes.c_uint32)]


class DgDrawCmd(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("index", ctypes.c_uint32)]


class DgFrameLists(ctypes.Structure):
    _fields_ = [("view", DgView),
                ("renders", ctypes.POINTER(DgBitmapRender)), ("n_renders", ctypes.c_uint32),
                ("columns", ctypes.POINTER(DgBitmapColumn)), ("n_columns", ctypes.c_uint32),
                ("visplanes", ctypes.POINTER(DgVisplane)), ("n_visplanes", ctypes.c_uint32),
                ("plane_tb", ctypes.POINTER(ctypes.c_int16)), ("n_plane_tb", ctypes.c_uint32),
                ("order", ctypes.POINTER(DgDrawCmd)), ("n_order", ctypes.c_uint32)]


def build(force: bool = False) -> str:
    """Compile libdoomgpu.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [INCLUDE]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", csrc, "-s"])
    return LIB_PATH


_lib = None

# every symbol include/doomgpu.h declares: (restype, argtypes)
_P = ctypes.c_void_p
_SIGNATURES = {
    "dg_scene_load_wad": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.POINTER(_P)]),
    "dg_scene_free": (None, [_P]),
    "dg_scene_player_start": (ctypes.c_int, [_P] + [ctypes.POINTER(ctypes.c_float)] * 3),
    "dg_scene_floor_height_at": (ctypes.c_int, [_P, ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_float)]),
    "dg_scene_sector_count": (ctypes.c_int, [_P]),
    "dg_scene_set_sector_light": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int16]),
    "dg_scene_mobj_count": (ctypes.c_int, [_P]),
    "dg_scene_set_mobj_state": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint8, ctypes.c_int]),
    "dg_create": (ctypes.c_int, [ctypes.POINTER(DgConfig), ctypes.POINTER(_P)]),
    "dg_destroy": (None, [_P]),
    "dg_ctx_host_threads": (ctypes.c_int, [_P]),
    "dg_upload_scene": (ctypes.c_int, [_P, _P]),
    "dg_render_views": (ctypes.c_int, [_P, ctypes.POINTER(DgView), ctypes.c_int, _P]),
    "dg_submit_views": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(DgView), ctypes.c_int]),
    "dg_submit_views_state": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(DgView), ctypes.POINTER(DgViewState), ctypes.c_int]),
    "dg_render_views_state": (ctypes.c_int, [_P, ctypes.POINTER(DgView), ctypes.POINTER(DgViewState), ctypes.c_int, _P]),
    "dg_scene_sprite_frame": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_uint8]),
    "dg_wait": (ctypes.c_int, [_P, ctypes.c_int]),
    "dg_slot_framebuffer": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(_P)]),
    "dg_readback": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "dg_readback_async": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    "dg_ctx_fallbacks": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64)]),
    "dg_ctx_redone_frames": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64)]),
    "dg_frame_checksums": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    "dg_alloc_host": (_P, [ctypes.c_size_t]),
    "dg_free_host": (None, [_P]),
    "dg_prepare_views": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(DgView), ctypes.c_int]),
    "dg_replay_slot": (ctypes.c_int, [_P, ctypes.c_int]),
    "dg_scene_texture_id": (ctypes.c_int, [_P, ctypes.c_char_p]),
    "dg_scene_flat_id": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_float]),
    "dg_scene_sprite_bitmap_id": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_uint8, ctypes.c_uint8]),
    "dg_scene_bitmap_size": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "dg_draw_lists": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(DgFrameLists), ctypes.c_int, _P]),
    "dg_build_lists": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.POINTER(DgView), ctypes.POINTER(DgFrameLists)]),
    "dg_last_error": (ctypes.c_char_p, []),
    "dg_version": (ctypes.c_char_p, []),
    "dg_slot_timing": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(DgTiming)]),
}


def lib():
    """Load libdoomgpu.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DoomGpuError(DG_ERR_NO_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() — there is no fallback path")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc: int):
    if rc < 0:
        raise DoomGpuError(rc, lib().dg_last_error().decode(errors="replace"))
    return rc


def make_view_states(states):
    """states: one (lights, mobjs) pair per view; lights = [(sector, light_level), ...], mobjs = [(mobj, sprite_frame or -1, full_bright), ...].
    Returns (ctypes array of dg_view_state, keep-alive list)."""
    arr = (DgViewState * len(states))()
    keep = []
    for i, (lights, mobjs) in enumerate(states):
        la = (DgSectorLight * max(1, len(lights)))(*[DgSectorLight(int(s), int(l)) for s, l in lights])
        ma = (DgMobjState * max(1, len(mobjs)))(*[DgMobjState(int(m), int(sf), int(fb), 0) for m, sf, fb in mobjs])
        keep += [la, ma]
        arr[i] = DgViewState(la, len(lights), ma, len(mobjs))
    return arr, keep


def make_views(records, timestamp: float = 0.0):
    """camera_path records (n, 8) f32 [x, y, angle, cos, sin, cos(-a), sin(-a), floor] -> ctypes array of dg_view."""
    recs = np.asarray(records, dtype=np.float32).reshape(-1, 8)
    arr = (DgView * len(recs))()
    for i, r in enumerate(recs):
        arr[i] = DgView(float(r[0]), float(r[1]), float(r[2]), float(r[7]), float(r[3]), float(r[4]), float(r[5]), float(r[6]),
                        float(timestamp), 1)
    return arr


class Scene:
    """dg_scene: Map + Palette + Textures + Flats + Sprites + MapObjects of one map (src/game.rs:142-167)."""

    def __init__(self, wad: bytes, map_name: str = "e1m1"):
        h = _P()
        _check(lib().dg_scene_load_wad(wad, len(wad), map_name.encode(), ctypes.byref(h)))
        self._h = h

    def player_start(self):
        x, y, a = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        _check(lib().dg_scene_player_start(self._h, x, y, a))
        return x.value, y.value, a.value

    def floor_height_at(self, x: float, y: float, default: float = 0.0) -> float:
        h = ctypes.c_float(default)
        _check(lib().dg_scene_floor_height_at(self._h, x, y, h))
        return h.value

    def sector_count(self) -> int:
        return lib().dg_scene_sector_count(self._h)

    def set_sector_light(self, sector: int, light: int):
        _check(lib().dg_scene_set_sector_light(self._h, sector, light))

    def mobj_count(self) -> int:
        return lib().dg_scene_mobj_count(self._h)

    def sprite_frame(self, sprite: str, frame: int = 0) -> int:
        """Handle for dg_mobj_state.sprite_frame (may decode bitmaps: call before Context.upload_scene)."""
        return _check(lib().dg_scene_sprite_frame(self._h, sprite.encode(), frame))

    def set_mobj_state(self, mobj: int, sprite, frame: int = 0, full_bright: bool = False):
        _check(lib().dg_scene_set_mobj_state(self._h, mobj, sprite.encode() if sprite else None, frame, int(full_bright)))

    def build_lists(self, W: int, H: int, view: DgView) -> DgFrameLists:
        fl = DgFrameLists()
        _check(lib().dg_build_lists(self._h, W, H, ctypes.byref(view), ctypes.byref(fl)))
        return fl

    def close(self):
        if self._h:
            lib().dg_scene_free(self._h)
            self._h = None


class Context:
    """dg_ctx on one GPU.  Mirrors the reference call shape: for every view, `Pixels::new()` +
    `Renderer::new(..).render()` -> `pixels.pixels` (src/game.rs:505-525), batched."""

    def __init__(self, width: int, height: int, max_batch: int = 64, slots: int = 2, device: int = 0, host_threads: int = 0,
                 front_end: int = 0):
        cfg = DgConfig(device, width, height, max_batch, slots, host_threads, front_end)
        h = _P()
        _check(lib().dg_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self.width, self.height, self.max_batch, self.slots = width, height, max_batch, slots
        self.frame_bytes = 3 * width * height
        self.host_threads = lib().dg_ctx_host_threads(self._h)
        self._scene = None

    def upload_scene(self, scene: Scene):
        _check(lib().dg_upload_scene(self._h, scene._h))
        self._scene = scene  # keep alive

    def render(self, views) -> np.ndarray:
        """Synchronous full path; returns (n, H, W, 3) uint8."""
        n = len(views)
        out = np.empty((n, self.height, self.width, 3), dtype=np.uint8)
        _check(lib().dg_render_views(self._h, views, n, out.ctypes.data_as(_P)))
        return out

    def render_one_into(self, view: DgView, host_ptr: int):
        """`Renderer::new(..).render()` for ONE view, synchronously, RGB24 written to caller memory (the drop-in call shape:
        rust/src/gpu.rs GpuRenderer::render)."""
        _check(lib().dg_render_views(self._h, ctypes.byref(view), 1, _P(host_ptr)))

    def submit(self, slot: int, views, n=None, states=None):
        if states is None:
            _check(lib().dg_submit_views(self._h, slot, views, len(views) if n is None else n))
        else:
            _check(lib().dg_submit_views_state(self._h, slot, views, states, len(views) if n is None else n))

    def render_state(self, views, states) -> np.ndarray:
        """render() with one game-state snapshot per view (make_view_states)."""
        n = len(views)
        out = np.empty((n, self.height, self.width, 3), dtype=np.uint8)
        _check(lib().dg_render_views_state(self._h, views, states, n, out.ctypes.data_as(_P)))
        return out

    def wait(self, slot: int):
        _check(lib().dg_wait(self._h, slot))

    def prepare(self, slot: int, views):
        _check(lib().dg_prepare_views(self._h, slot, views, len(views)))

    def replay(self, slot: int):
        _check(lib().dg_replay_slot(self._h, slot))

    def readback(self, slot: int, first: int, count: int) -> np.ndarray:
        out = np.empty((count, self.height, self.width, 3), dtype=np.uint8)
        _check(lib().dg_readback(self._h, slot, first, count, out.ctypes.data_as(_P)))
        return out

    def frame_checksums(self, slot: int, first: int, count: int) -> np.ndarray:
        """One uint64 per frame, computed on the GPU (dg_frame_checksums); `frame_checksum` is the host-side twin."""
        out = np.zeros(count, dtype=np.uint64)
        _check(lib().dg_frame_checksums(self._h, slot, first, count, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return out

    def readback_async(self, slot: int, first: int, count: int, host_ptr: int):
        """Queue the D2H copy behind the slot's kernels (own copy stream); complete after wait(slot)."""
        _check(lib().dg_readback_async(self._h, slot, first, count, _P(host_ptr)))

    def fallbacks(self) -> dict:
        a, f = ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib().dg_ctx_fallbacks(self._h, ctypes.byref(a)))
        _check(lib().dg_ctx_redone_frames(self._h, ctypes.byref(f)))
        return {"front_end": a.value, "redone_frames": f.value}

    def readback_into(self, slot: int, first: int, count: int, host_ptr: int):
        _check(lib().dg_readback(self._h, slot, first, count, _P(host_ptr)))

    def framebuffer_ptr(self, slot: int) -> int:
        p = _P()
        _check(lib().dg_slot_framebuffer(self._h, slot, ctypes.byref(p)))
        return p.value

    def draw_lists(self, slot: int, frames) -> np.ndarray:
        n = len(frames)
        out = np.empty((n, self.height, self.width, 3), dtype=np.uint8)
        _check(lib().dg_draw_lists(self._h, slot, frames, n, out.ctypes.data_as(_P)))
        return out

    def timing(self, slot: int) -> dict:
        t = DgTiming()
        _check(lib().dg_slot_timing(self._h, slot, ctypes.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_}

    def close(self):
        if self._h:
            lib().dg_destroy(self._h)
            self._h = None


def frame_checksum(rgb24) -> int:
    """dg_frame_checksums' formula on the host, for one frame given as bytes / uint8 array of length 3*W*H (a byte count that is not a
    multiple of 4 ends in a zero-extended partial dword)."""
    raw = bytes(rgb24) if not isinstance(rgb24, np.ndarray) else np.ascontiguousarray(rgb24).reshape(-1).tobytes()
    raw += b"\0" * (-len(raw) % 4)
    d = np.frombuffer(raw, dtype="<u4")
    with np.errstate(over="ignore"):
        i = np.arange(d.size, dtype=np.uint64)
        m = (d.astype(np.uint64) ^ (i * np.uint64(0x9E3779B97F4A7C15))) * np.uint64(0xBF58476D1CE4E5B9)
        return int((m ^ (m >> np.uint64(32))).sum(dtype=np.uint64))


def declared_symbols() -> list[str]:
    """Function names declared in include/doomgpu.h (for the export check)."""
    import re
    txt = open(INCLUDE).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z_0-9]+)\s*\(", txt)))
