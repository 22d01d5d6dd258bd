#!/usr/bin/env python3
"""compare_ref.py — compare frames dumped by the UNMODIFIED Rust reference (tools/ref_dump/main_dump.rs) with libdoomgpu on
the GPU and with the CPU oracle.  This is what turns "parity unpinned" into a pinned statement; it needs a machine that can
run the reference (cargo + libSDL2 + a WAD) for the dump, and an MI355X for the GPU leg (skipped with --no-gpu).

    python tools/compare_ref.py --wad doom1.wad --map e1m1 --size 320x200 --views views.txt --frames frames.rgb [--trig dump.log] [--no-gpu]
    python tools/compare_ref.py --emit-views views.txt [--wad doom1.wad --map e1m1 | --synth-seed 1993] [--frames-per-path 1000] [--stride 10] [--start]

views.txt: `x y angle` per line (%.9g: round-trips f32 exactly), or the word `start` (the Player-1 start as Game::new takes it,
src/game.rs:151-156).  frames.rgb: W*H*3 bytes per view, back to back.  dump.log: the dumper's stdout (position, angle, cos/sin bits).
--emit-views writes the camera path the benchmark and the golden fixtures use (camera_path.make_camera_path over synth_route for the
synthetic IWAD, route_from_wad for a real one) in the dumper's format, so the reference renders exactly the committed views.
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


def emit_views(a, cp):
    """The committed camera path as a views file for tools/ref_dump/main_dump.rs."""
    sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
    if a.synth_seed is not None:
        wad = sw.build_synth_iwad(a.synth_seed)
        route = sw.synth_route(a.synth_seed)
        with open(a.emit_views + ".wad", "wb") as f:                     # the IWAD the reference has to be run on (map E1M1)
            f.write(wad)
    else:
        if not a.wad:
            raise SystemExit("--emit-views needs --wad or --synth-seed")
        wad = open(a.wad, "rb").read()
        route = cp.route_from_wad(wad, a.map)
    import doomref
    osc = doomref.Scene(wad, a.map)
    recs = cp.make_camera_path(route, osc.floor_height_at, a.frames_per_path)[:: max(1, a.stride)]
    with open(a.emit_views, "w") as f:
        if a.start:
            f.write("start\n")
        for r in recs:
            f.write("%.9g %.9g %.9g\n" % (r[0], r[1], r[2]))
    print(f"{a.emit_views}: {len(recs) + (1 if a.start else 0)} views" + (f"; IWAD written to {a.emit_views}.wad" if a.synth_seed is not None else ""))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wad", default=None)
    ap.add_argument("--map", default="e1m1")
    ap.add_argument("--size", default="320x200")
    ap.add_argument("--views", default=None)
    ap.add_argument("--frames", default=None)
    ap.add_argument("--emit-views", default=None, metavar="OUT", help="write a views file for the dumper instead of comparing")
    ap.add_argument("--synth-seed", type=int, default=None, help="--emit-views: the synthetic IWAD of this seed instead of --wad (also writes OUT.wad)")
    ap.add_argument("--frames-per-path", type=int, default=1000)
    ap.add_argument("--stride", type=int, default=1, help="--emit-views: every n-th frame of the path")
    ap.add_argument("--start", action="store_true", help="--emit-views: first line `start` (Player-1 start)")
    ap.add_argument("--trig", default=None, help="stdout of the dumper: cos/sin(+-angle) bit patterns of the dumping machine's libm")
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    cp = importlib.import_module("doom-rust-renderer_amd.camera_path")
    if a.emit_views:
        return emit_views(a, cp)
    if not (a.wad and a.views and a.frames):
        ap.error("--wad, --views and --frames are required unless --emit-views is given")
    W, H = (int(t) for t in a.size.lower().split("x"))
    wad = open(a.wad, "rb").read()
    import doomref
    osc = doomref.Scene(wad, a.map)
    lines = [l.split() for l in open(a.views)]
    lines = [l for l in lines if l[:1] == ["start"] or len(l) >= 3]
    trig = None
    dumped = None
    if a.trig:
        logl = [l for l in open(a.trig) if "trig" in l]
        trig = [[np.array([int(h, 16)], dtype="<u4").view("<f4")[0] for h in l.split("trig")[1].split()[:4]] for l in logl]
        dumped = [[float(t) for t in l.split()[:3]] for l in logl]       # the position / angle the dumper actually used
        assert len(trig) == len(lines), "dump log and views file disagree"
    views = []
    for i, l in enumerate(lines):
        if l[0] == "start":                                              # src/game.rs:151-156; things.rs:36 converts degrees with f32::to_radians
            views.append(dumped[i] if dumped else list(osc.player_start())[:3])
        else:
            views.append([float(t) for t in l[:3]])
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
