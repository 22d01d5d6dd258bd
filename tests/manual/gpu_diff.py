"""Where does the GPU frame differ from the oracle?  python3 tests/manual/gpu_diff.py [W H frame ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import doomref
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
path = np.fromfile(os.path.join(ROOT, "tests/golden/campath_seed1993.f32"), dtype="<f4").reshape(1000, 8)
wad = sw.build_synth_iwad(1993)
osc = doomref.Scene(wad, "e1m1"); sc = dg.Scene(wad, "e1m1")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 200)
frames = [int(a) for a in sys.argv[3:]] or [0, 100, 297, 728]
ctx = dg.Context(W, H, max_batch=len(frames), slots=1); ctx.upload_scene(sc)
out = ctx.render(dg.make_views(path[frames]))
for k, i in enumerate(frames):
    ref = np.frombuffer(osc.render(W, H, path[i]), dtype=np.uint8).reshape(H, W, 3)
    bad = np.any(out[k] != ref, axis=2)
    ys, xs = np.nonzero(bad)
    print(f"frame {i}: {bad.sum()} differing pixels of {W*H}")
    if bad.sum():
        print(f"  rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}; columns affected {len(set(xs))}; rows affected {len(set(ys))}")
        for (y, x) in list(zip(ys, xs))[:8]:
            print(f"  ({x},{y}) gpu {out[k][y, x]} ref {ref[y, x]}")
        cols = sorted(set(xs)); print("  first columns:", cols[:20]); print("  lane (x%64) histogram:", np.bincount(np.array(xs) % 64, minlength=64).tolist())
        print("  row%4 histogram:", np.bincount(np.array(ys) % 4, minlength=4).tolist(), " row-in-band(25/50) hist:", np.bincount(np.array(ys) % (25 if H == 200 else 50))[:50].tolist())
