#!/usr/bin/env python3
"""compare_ref.py — compare frames dumped by the UNMODIFIED Rust reference (tools/ref_dump/main_dump.rs) with libdoomgpu on
the GPU and with the CPU oracle.  This is what turns "parity unpinned" into a pinned statement; it needs a machine that can
run the reference (cargo + libSDL2 + a WAD) for the dump, and an MI355X for the GPU leg (skipped with --no-gpu).

    python tools/compare_ref.py --wad doom1.wad --map e1m1 --size 320x200 --views views.txt --frames frames.rgb [--trig dump.log] [--no-gpu]

views.txt: `x y angle` per line.  frames.rgb: W*H*3 bytes per view, back to back.  dump.log: the dumper's stdout (cos/sin bits).
Exit status 0 iff every compared frame is byte-identical.
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def first_diff(a: np.ndarray, b: np.ndarray):
    bad = np.argwhere(np.any(a != b, axis=2))
    return None if len(bad) == 0 else (int(bad[0][1]), int(bad[0][0]), a[bad[0][0], bad[0][1]].tolist(), b[bad[0][0], bad[0][1]].tolist(), len(bad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wad", required=True)
    ap.add_argument("--map", default="e1m1")
    ap.add_argument("--size", default="320x200")
    ap.add_argument("--views", required=True)
    ap.add_argument("--frames", required=True)
    ap.add_argument("--trig", default=None, help="stdout of the dumper: cos/sin(+-angle) bit patterns of the dumping machine's libm")
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    W, H = (int(t) for t in a.size.lower().split("x"))
    wad = open(a.wad, "rb").read()
    cp = importlib.import_module("doom-rust-renderer_amd.camera_path")
    import doomref
    osc = doomref.Scene(wad, a.map)
    views = [[float(t) for t in l.split()[:3]] for l in open(a.views) if len(l.split()) >= 3]
    trig = None
    if a.trig:
        trig = [[np.array([int(h, 16)], dtype="<u4").view("<f4")[0] for h in l.split("trig")[1].split()[:4]] for l in open(a.trig) if "trig" in l]
        assert len(trig) == len(views), "dump log and views file disagree"
    raw = np.fromfile(a.frames, dtype=np.uint8)
    assert raw.size == len(views) * W * H * 3, f"{a.frames}: expected {len(views)} frames of {W}x{H}"
    ref = raw.reshape(len(views), H, W, 3)

    dg = None if a.no_gpu else importlib.import_module("doom-rust-renderer_amd")
    recs = []
    scene = dg.Scene(wad, a.map) if dg else None
    for i, (x, y, ang) in enumerate(views):
        fh = 0.0
        if scene is not None:
            fh = scene.floor_height_at(x, y, 0.0)
        else:
            fh = osc.floor_height_at(x, y) if hasattr(osc, "floor_height_at") else 0.0
        r = cp.view_record(x, y, ang, fh)
        if trig:
            r[3:7] = trig[i]
        recs.append(r)
    recs = np.array(recs, dtype=np.float32)
    ok = True
    for i in range(len(views)):
        o = np.frombuffer(osc.render(W, H, recs[i]), dtype=np.uint8).reshape(H, W, 3)
        d = first_diff(o, ref[i])
        print(f"view {i}: oracle vs reference: " + ("identical" if d is None else f"{d[4]} pixels differ, first at ({d[0]},{d[1]}): oracle {d[2]} reference {d[3]}"))
        ok &= d is None
    if dg:
        ctx = dg.Context(W, H, max_batch=len(views), slots=1)
        ctx.upload_scene(scene)
        out = ctx.render(dg.make_views(recs))
        sums = ctx.frame_checksums(0, 0, len(views))
        for i in range(len(views)):
            d = first_diff(out[i], ref[i])
            same_sum = int(sums[i]) == dg.frame_checksum(ref[i])
            print(f"view {i}: GPU vs reference: " + ("identical" if d is None else f"{d[4]} pixels differ, first at ({d[0]},{d[1]}): gpu {d[2]} reference {d[3]}") +
                  f"; dg_frame_checksums {'matches' if same_sum else 'differs'}")
            ok &= d is None and same_sum
        ctx.close()
    print("PARITY PINNED for these views" if ok else "DIFFERENCES FOUND")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
