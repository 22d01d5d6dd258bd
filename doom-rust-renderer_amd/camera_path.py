"""Scripted camera path (SURVEY.md §8d "Camera path").

A closed Catmull-Rom loop (evaluated in float64, sampled uniformly in the spline parameter) through
the waypoints of ``synth_wad.synth_route``; heading = path tangent + 0.35*sin(2*pi*i/125).  Each frame is
stored as eight f32: x, y, angle, cos(angle), sin(angle), cos(-angle), sin(-angle), floor_height —
the reference's `Player` (src/game.rs:40-45) plus the trig values `Vertex::rotate` would obtain from
libm (src/map/vertexes.rs:20-25), recorded so that CPU and GPU consume identical bits.
cos/sin come from the C library's cosf/sinf (what Rust's f32::cos/sin call), not numpy's SIMD kernels.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _f in ("cosf", "sinf"):
    getattr(_libm, _f).restype = ctypes.c_float
    getattr(_libm, _f).argtypes = [ctypes.c_float]


def view_record(x: float, y: float, angle: float, floor_height: float) -> np.ndarray:
    a = np.float32(angle)
    na = np.float32(-a)
    return np.array([x, y, a, _libm.cosf(a), _libm.sinf(a), _libm.cosf(na), _libm.sinf(na), floor_height], dtype=np.float32)


def catmull_rom_loop(points, n_frames: int) -> np.ndarray:
    """(n_frames, 3) float64: x, y, tangent angle."""
    p = np.asarray(points, dtype=np.float64)
    n = len(p)
    out = np.zeros((n_frames, 3))
    for i in range(n_frames):
        u = i * n / n_frames
        k = int(math.floor(u))
        t = u - k
        p0, p1, p2, p3 = p[(k - 1) % n], p[k % n], p[(k + 1) % n], p[(k + 2) % n]
        pos = 0.5 * ((2 * p1) + (-p0 + p2) * t + (2 * p0 - 5 * p1 + 4 * p2 - p3) * t * t + (-p0 + 3 * p1 - 3 * p2 + p3) * t ** 3)
        tan = 0.5 * ((-p0 + p2) + 2 * (2 * p0 - 5 * p1 + 4 * p2 - p3) * t + 3 * (-p0 + 3 * p1 - 3 * p2 + p3) * t * t)
        ang = math.atan2(tan[1], tan[0]) if (abs(tan[0]) + abs(tan[1])) > 1e-9 else 0.0
        out[i] = (pos[0], pos[1], ang)
    return out


def make_camera_path(route, floor_height_fn, n_frames: int = 1000) -> np.ndarray:
    """(n_frames, 8) float32 view records.  floor_height_fn(x, y) -> sector floor height at the eye
    (the reference's update_current_player_height, src/game.rs:376-389)."""
    xyz = catmull_rom_loop(route, n_frames)
    recs = np.zeros((n_frames, 8), dtype=np.float32)
    prev = 0.0
    for i in range(n_frames):
        x, y = np.float32(xyz[i, 0]), np.float32(xyz[i, 1])
        ang = xyz[i, 2] + 0.35 * math.sin(2.0 * math.pi * i / 125.0)
        fh = floor_height_fn(float(x), float(y), prev)
        prev = fh
        recs[i] = view_record(x, y, ang, fh)
    return recs


def route_from_wad(wad: bytes, map_name: str, max_points: int = 48):
    """A closed walk for ANY map (bench.py --wad): the midpoints of its two-sided linedefs (doorways, steps, windows) sorted by
    angle around their centroid, thinned to max_points, starting nearest to the Player-1 start.  Only a benchmark path: it may
    cut through walls or the void, which the renderer handles like the reference does."""
    import struct
    n, off = struct.unpack_from("<II", wad, 4)
    lumps = []
    for i in range(n):
        o, sz = struct.unpack_from("<II", wad, off + 16 * i)
        lumps.append((wad[off + 16 * i + 8: off + 16 * i + 16].split(b"\0")[0].decode("ascii", "replace").upper(), o, sz))
    idx = [i for i, (nm, _, _) in enumerate(lumps) if nm == map_name.upper()]
    if not idx:
        raise ValueError(f"map {map_name} not in WAD")
    m = idx[0]                                         # map lumps follow the FIRST marker of that name (src/wad.rs:175-183)
    def lump(k):
        _, o, sz = lumps[m + k]
        return wad[o:o + sz]
    things, linedefs, vertexes = lump(1), lump(2), lump(4)
    vx = np.frombuffer(vertexes, dtype="<i2").reshape(-1, 2).astype(np.float64)
    pts = []
    for i in range(len(linedefs) // 14):
        v1, v2, _flags, _, _, _front, back = struct.unpack_from("<hhhhhhh", linedefs, i * 14)
        if back >= 0 and 0 <= v1 < len(vx) and 0 <= v2 < len(vx):
            pts.append((vx[v1] + vx[v2]) / 2.0)
    if len(pts) < 4:                                   # hardly any openings: walk the vertex cloud's bounding box instead
        lo, hi = vx.min(axis=0), vx.max(axis=0)
        c, r = (lo + hi) / 2.0, (hi - lo) / 4.0
        pts = [c + r * np.array([math.cos(t), math.sin(t)]) for t in np.linspace(0, 2 * math.pi, 12, endpoint=False)]
    pts = np.array(pts)
    c = pts.mean(axis=0)
    pts = pts[np.argsort(np.arctan2(pts[:, 1] - c[1], pts[:, 0] - c[0]), kind="stable")]
    if len(pts) > max_points:
        pts = pts[np.linspace(0, len(pts), max_points, endpoint=False).astype(int)]
    start = None
    for i in range(len(things) // 10):
        x, y, _a, t, _f = struct.unpack_from("<hhhhh", things, i * 10)
        if t == 1:
            start = np.array([x, y], dtype=np.float64)
    if start is not None:
        k = int(np.argmin(((pts - start) ** 2).sum(axis=1)))
        pts = np.concatenate([pts[k:], pts[:k]])
    return [(float(x), float(y)) for x, y in pts]
