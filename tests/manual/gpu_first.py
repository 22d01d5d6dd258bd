"""First GPU contact: parity of the HIP path vs the oracle on sampled path frames + kernel timings."""
import importlib, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import doomref
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
cp = importlib.import_module("doom-rust-renderer_amd.camera_path")

seed = 1993
wad = sw.build_synth_iwad(seed)
osc = doomref.Scene(wad, "e1m1")
sc = dg.Scene(wad, "e1m1")
path = cp.make_camera_path(sw.synth_route(seed), lambda x, y, d: sc.floor_height_at(x, y, d), 1000)
print("version", dg.lib().dg_version())
for (W, H, B, stride) in [(320, 200, 64, 16), (1280, 800, 16, 64), (1024, 768, 8, 125), (2560, 1600, 4, 250)]:
    ctx = dg.Context(W, H, max_batch=B, slots=2)
    ctx.upload_scene(sc)
    idx = list(range(0, 1000, stride))[:B]
    views = dg.make_views(path[idx])
    t = time.time(); out = ctx.render(views); dt = time.time() - t
    bad = 0
    for k, i in enumerate(idx):
        ref = np.frombuffer(osc.render(W, H, path[i]), dtype=np.uint8).reshape(H, W, 3)
        if not np.array_equal(ref, out[k]):
            bad += 1
            d = np.nonzero((ref != out[k]).any(axis=2))
            print("  frame", i, "differs at", len(d[0]), "pixels, first", list(zip(d[0][:4], d[1][:4])), "ref", ref[d[0][0], d[1][0]], "gpu", out[k][d[0][0], d[1][0]])
    print(f"{W}x{H}: {len(idx)} frames, mismatching {bad}, first render {dt*1e3:.1f} ms, timing {ctx.timing(0)}")
    # steady-state kernel timing
    ctx.prepare(0, views)
    for _ in range(3): ctx.replay(0); ctx.wait(0)
    ts = []
    for _ in range(10):
        ctx.replay(0); ctx.wait(0); ts.append(ctx.timing(0))
    rm = np.median([t["raster_ms"] for t in ts]); sm = np.median([t["setup_ms"] for t in ts])
    nf = len(idx)
    print(f"   replay: setup {sm:.3f} ms raster {rm:.3f} ms for {nf} frames -> {nf/((rm+sm)/1e3):.0f} fps kernel-only; "
          f"alg bytes/frame {4*W*H} -> {4*W*H*nf/(rm/1e3)/1e9:.1f} GB/s")
    ctx.close()
