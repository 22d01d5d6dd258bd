"""Debug aid: render given path frames through libdoomgpu and the oracle, print where they differ.
    python tests/manual/gpu_diff_frame.py [front_end 1|2] [W H] frame ..."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import doomref
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
fe, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frames = [int(a) for a in sys.argv[4:]]
path = np.fromfile(os.path.join(ROOT, "tests/golden/campath_seed1993.f32"), dtype="<f4").reshape(1000, 8)
wad = sw.build_synth_iwad(1993)
osc = doomref.Scene(wad, "e1m1"); sc = dg.Scene(wad, "e1m1")
ctx = dg.Context(W, H, max_batch=max(250, len(frames)), slots=2, front_end=fe); ctx.upload_scene(sc)
lo = min(frames)
out = ctx.render(dg.make_views(path[lo:lo + 250]))
for i in frames:
    ref = np.frombuffer(osc.render(W, H, path[i]), dtype=np.uint8).reshape(H, W, 3)
    bad = np.argwhere(np.any(out[i - lo] != ref, axis=2))
    print(f"frame {i}: {len(bad)} pixels differ")
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        print("  columns", sorted(set(xs.tolist()))[:40], "rows", ys.min(), "..", ys.max())
        for (y, x) in bad[:12]:
            print(f"  ({x},{y}) gpu {out[i - lo][y, x].tolist()} ref {ref[y, x].tolist()}")
