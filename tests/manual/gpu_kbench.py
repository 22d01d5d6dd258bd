"""Kernel tuning loop: parity on sampled frames, then median replay time of 250 resident frames at 1280x800 (and 320x200)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import doomref
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
path = np.fromfile(os.path.join(ROOT, "tests/golden/campath_seed1993.f32"), dtype="<f4").reshape(1000, 8)
wad = sw.build_synth_iwad(1993)
osc = doomref.Scene(wad, "e1m1"); sc = dg.Scene(wad, "e1m1")
SIZES = [tuple(int(v) for v in t.split("x")) for t in os.environ.get("KBENCH_SIZES", "1280x800x250,320x200x1000").split(",")]
for (W, H, B) in SIZES:
    ctx = dg.Context(W, H, max_batch=B, slots=1); ctx.upload_scene(sc)
    idx = [0, 100, 297, 323, 623, 728]
    out = ctx.render(dg.make_views(path[idx]))
    bad = sum(not np.array_equal(out[k], np.frombuffer(osc.render(W, H, path[i]), dtype=np.uint8).reshape(H, W, 3)) for k, i in enumerate(idx))
    ctx.prepare(0, dg.make_views(path[:B]))
    for _ in range(3): ctx.replay(0); ctx.wait(0)
    ts = []
    for _ in range(15):
        ctx.replay(0); ctx.wait(0); ts.append(ctx.timing(0))
    rm = float(np.median([t["raster_ms"] for t in ts])); sm = float(np.median([t["setup_ms"] for t in ts]))
    t = ts[-1]
    alg = B * (4 * W * H + 4 * (W + 1)) + 24 * t["n_spans"] + 48 * t["n_walls"] + 16 * t["n_planes"]
    print(f"{W}x{H} B={B}: mismatches {bad}/{len(idx)} | setup {sm:.3f} ms raster {rm:.3f} ms | {B/((rm+sm)/1e3):.0f} fps | "
          f"{alg/(rm/1e3)/1e9:.0f} GB/s alg = {alg/(rm/1e3)/8e12*100:.1f}% of 8 TB/s | spans/frame {t['n_spans']/B:.0f}")
    ctx.close()
