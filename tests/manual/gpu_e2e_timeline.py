"""Where does the end-to-end time go?  Per-call wall times of submit / wait on alternating slots (1280x800, 250 frames)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
path = np.fromfile(os.path.join(ROOT, "tests/golden/campath_seed1993.f32"), dtype="<f4").reshape(1000, 8)
sc = dg.Scene(sw.build_synth_iwad(1993), "e1m1")
W, H, B = 1280, 800, 250
for fe in (dg.DG_FE_HOST, dg.DG_FE_DEVICE):
    for slots in (2, 4):
        for threads in (0, 8):
            ctx = dg.Context(W, H, max_batch=B, slots=slots, front_end=fe, host_threads=threads); ctx.upload_scene(sc)
            views = [dg.make_views(path[i * B:(i + 1) * B]) for i in range(4)]
            for k in range(2 * slots): ctx.submit(k % slots, views[k % 4])
            for k in range(slots): ctx.wait(k)
            calls = []
            t0 = time.perf_counter()
            N = 24
            for k in range(N):
                a = time.perf_counter(); ctx.submit(k % slots, views[k % 4]); calls.append(time.perf_counter() - a)
            for k in range(slots): ctx.wait(k)
            dt = time.perf_counter() - t0
            t = ctx.timing(0)
            print(f"fe {fe} slots {slots} threads {ctx.host_threads}: {N*B/dt:.0f} fps | per batch {dt/N*1e3:.3f} ms | submit call median {np.median(calls)*1e3:.3f} ms "
                  f"| host_ms {t['host_ms']:.3f} | gpu front {t['setup_ms']:.3f} raster {t['raster_ms']:.3f} total {t['total_ms']:.3f}", flush=True)
            ctx.close()
