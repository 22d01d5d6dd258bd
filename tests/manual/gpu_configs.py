"""Kernel-resident throughput of the BASELINE.json configurations that fit one GPU (synthetic stand-ins): table for DESIGN.md."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
print("| config | map | size | frames/launch | front end | front-end kernels ms | raster ms | frames/s (resident) | GB/s alg | frac | spans/frame | records/frame |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for (name, seed, W, H, B) in [("1 / 2", 1993, 320, 200, 1000), ("3", 1993, 1280, 800, 250), ("native", 1993, 1024, 768, 250), ("4 (second map)", 1994, 1280, 800, 250),
                              ("5", 1994, 2560, 1600, 64), ("3 @ 2560x1600", 1993, 2560, 1600, 64), ("vanilla-shaped", 1995, 1280, 800, 250)]:
    path = np.fromfile(os.path.join(ROOT, f"tests/golden/campath_seed{seed}.f32"), dtype="<f4").reshape(1000, 8)
    sc = dg.Scene(sw.build_synth_iwad(seed, heavy=(seed == 1994), vanilla=(seed == 1995)), "e1m1")
    for fe in (dg.DG_FE_DEVICE, dg.DG_FE_HOST):
        ctx = dg.Context(W, H, max_batch=B, slots=1, front_end=fe)
        ctx.upload_scene(sc)
        ctx.prepare(0, dg.make_views(path[:B]))
        for _ in range(3):
            ctx.replay(0); ctx.wait(0)
        ts = []
        for _ in range(15):
            ctx.replay(0); ctx.wait(0); ts.append(ctx.timing(0))
        rm = float(np.median([t["raster_ms"] for t in ts])); sm = float(np.median([t["setup_ms"] for t in ts])); t = ts[-1]
        alg = B * (4 * W * H + 4 * (W + 1)) + 32 * t["n_spans"]
        print(f"| {name} | seed {seed}{' (heavy)' if seed == 1994 else ''} | {W}x{H} | {B} | {'device' if t['front_end'] == 2 else 'host lists'} | {sm:.3f} | {rm:.3f} | "
              f"{B/((rm+sm)/1e3):,.0f} | {alg/(rm/1e3)/1e9:,.0f} | {alg/(rm/1e3)/8e12*100:.1f} % | {t['n_spans']/B:.0f} | {t['n_walls']/B:.0f} |", flush=True)
        ctx.close()
    sc.close()
