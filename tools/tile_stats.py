#!/usr/bin/env python3
"""tile_stats.py — what dg_raster_tiles' loops meet on the benchmark path (CPU only, through tests/emul): chunks, owner-loop trips,
overlay evaluations, mixed-kind chunks, staged spans ...  A tuning aid: the numbers in profiles/r03_raster_tiles.md come from it.

    python tools/tile_stats.py [--size 1280x800] [--seed 1993] [--heavy] [--stride 50]
"""
import argparse, ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emul_bind

NAMES = ["chunks", "opaque span hits", "overlay span hits", "overlay evaluations", "single-kind chunks", "flat+wall chunks", "single-owner chunks",
         "staged spans (sum over tiles)", "staged spans touching their tile", "flat px", "wall px", "sky px", "uncovered px", "tiles", "tiles with overlays",
         "columns > 8 spans (per tile)", "flat chunks", "flat chunks = same planes as 8 columns left",
         "no-span chunks", "sole-owner wall chunks", "sole-owner flat chunks", "(unused)", "general-path chunks", "tiles without a general-path chunk",
         "2 owners, no overlay", "2 owners + overlay", ">= 3 owners", "1 owner + overlay", "some row uncovered", "", "", ""]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1280x800"); ap.add_argument("--seed", type=int, default=1993); ap.add_argument("--heavy", action="store_true")
    ap.add_argument("--stride", type=int, default=50)
    a = ap.parse_args()
    W, H = map(int, a.size.split("x"))
    sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
    cp = importlib.import_module("doom-rust-renderer_amd.camera_path")
    wad = sw.build_synth_iwad(a.seed, heavy=a.heavy)
    L = emul_bind.lib()
    L.emul_tile_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(emul_bind.DgView), ctypes.POINTER(ctypes.c_uint64)]
    sc = emul_bind.EmulScene(wad)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import doomref
    osc = doomref.Scene(wad, "e1m1")
    path = cp.make_camera_path(sw.synth_route(a.seed, heavy=a.heavy), osc.floor_height_at, 1000)
    tot = np.zeros(32, dtype=np.float64); n = 0
    for i in range(0, 1000, a.stride):
        r = path[i]
        v = emul_bind.DgView(float(r[0]), float(r[1]), float(r[2]), float(r[7]), float(r[3]), float(r[4]), float(r[5]), float(r[6]), 0.0, 1)
        st = (ctypes.c_uint64 * 32)()
        assert L.emul_tile_stats(sc._h, W, H, ctypes.byref(v), st) == 0
        tot += np.array(list(st), dtype=np.float64); n += 1
    tot /= n
    print(f"{W}x{H}, seed {a.seed}{' heavy' if a.heavy else ''}, mean of {n} frames")
    for k, v in zip(NAMES, tot):
        print(f"  {k:48s} {v:12.1f}   per chunk {v / tot[0]:.3f}")
main()
