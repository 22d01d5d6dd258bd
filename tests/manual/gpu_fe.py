"""Device column walk vs host span lists: parity on sampled frames, kernel times of 250 resident frames, end-to-end submit rate."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import doomref
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
quick = "--quick" in sys.argv
for seed, heavy in [(1993, False), (1994, True)]:
    path = np.fromfile(os.path.join(ROOT, f"tests/golden/campath_seed{seed}.f32"), dtype="<f4").reshape(1000, 8)
    wad = sw.build_synth_iwad(seed, heavy=heavy)
    osc = doomref.Scene(wad, "e1m1"); sc = dg.Scene(wad, "e1m1")
    for (W, H, B) in [(320, 200, 100), (1280, 800, 250)]:
        res = {}
        for fe in (dg.DG_FE_HOST, dg.DG_FE_DEVICE):
            ctx = dg.Context(W, H, max_batch=B, slots=2, front_end=fe); ctx.upload_scene(sc)
            idx = [0, 100, 277, 297, 323, 623, 728, 809]
            out = ctx.render(dg.make_views(path[idx]))
            used = ctx.timing(0)["front_end"]
            bad = sum(not np.array_equal(out[k], np.frombuffer(osc.render(W, H, path[i]), dtype=np.uint8).reshape(H, W, 3)) for k, i in enumerate(idx))
            print(f"seed {seed} {W}x{H} front_end {fe} (used {used}): mismatches {bad}/{len(idx)}", flush=True)
            if bad or quick:
                ctx.close(); continue
            views = dg.make_views(path[:B])
            ctx.prepare(0, views)
            for _ in range(3): ctx.replay(0); ctx.wait(0)
            ts = []
            for _ in range(11):
                ctx.replay(0); ctx.wait(0); ts.append(ctx.timing(0))
            rm = float(np.median([t["raster_ms"] for t in ts])); sm = float(np.median([t["setup_ms"] for t in ts]))
            t = ts[-1]
            # end to end: submit alternating slots, wait
            for _ in range(2):
                ctx.submit(0, views); ctx.submit(1, views); ctx.wait(0); ctx.wait(1)
            t0 = time.perf_counter(); n = 0
            for _ in range(6):
                ctx.submit(0, views); ctx.submit(1, views); n += 2 * B
            ctx.wait(0); ctx.wait(1)
            dt = time.perf_counter() - t0
            print(f"   used {t['front_end']}: front-end kernels {sm:.3f} ms, raster {rm:.3f} ms per {B} frames -> {B/((rm+sm)/1e3):.0f} fps resident | "
                  f"host {t['host_ms']:.2f} ms/batch, H2D {t['list_bytes']/B/1024:.0f} KiB/frame, spans/frame {t['n_spans']/B:.0f} | e2e {n/dt:.0f} fps ({ctx.host_threads} threads)", flush=True)
            ctx.close()
