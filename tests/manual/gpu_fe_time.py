"""Front-end kernel time (dg_slot_timing.setup_ms) of one resident 250-frame batch at 1280x800, median of 15 replays."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
path = np.fromfile(os.path.join(ROOT, "tests/golden/campath_seed1993.f32"), dtype="<f4").reshape(1000, 8)
sc = dg.Scene(sw.build_synth_iwad(1993), "e1m1")
for (W, H, B) in [(1280, 800, 250), (320, 200, 1000)]:
    ctx = dg.Context(W, H, max_batch=B, slots=1); ctx.upload_scene(sc)
    ctx.prepare(0, dg.make_views(path[:B]))
    ts = []
    for _ in range(18):
        ctx.replay(0); ctx.wait(0); ts.append(ctx.timing(0)["setup_ms"])
    print(f"{W}x{H} B={B}: front-end kernels {np.median(ts[3:]) * 1e3:.1f} us (min {min(ts[3:]) * 1e3:.1f})")
    ctx.close()
