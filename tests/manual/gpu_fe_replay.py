"""Replay 250 resident frames at 1280x800 through the device column walk (for rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1993
W, H, B = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1280, 800, 250)
path = np.fromfile(os.path.join(ROOT, f"tests/golden/campath_seed{seed}.f32"), dtype="<f4").reshape(1000, 8)
sc = dg.Scene(sw.build_synth_iwad(seed, heavy=(seed == 1994)), "e1m1")
ctx = dg.Context(W, H, max_batch=B, slots=1, front_end=dg.DG_FE_DEVICE); ctx.upload_scene(sc)
ctx.prepare(0, dg.make_views(path[:B]))
ts = []
for _ in range(20):
    ctx.replay(0); ctx.wait(0); ts.append(ctx.timing(0))
print("front_end", ts[-1]["front_end"], "fe ms", np.median([t["setup_ms"] for t in ts]), "raster ms", np.median([t["raster_ms"] for t in ts]))
ctx.close()
