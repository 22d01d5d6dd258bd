"""Experiment: one 250-frame launch of the DG_EXP_T_TIMING build (device printf of per-wave phase cycles for a few tiles of frame 100)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
path = np.fromfile(os.path.join(ROOT, "tests/golden/campath_seed1993.f32"), dtype="<f4").reshape(1000, 8)
sc = dg.Scene(sw.build_synth_iwad(1993), "e1m1")
ctx = dg.Context(1280, 800, max_batch=250, slots=1); ctx.upload_scene(sc)
ctx.prepare(0, dg.make_views(path[:250]))
ctx.replay(0); ctx.wait(0)
ctx.close()
