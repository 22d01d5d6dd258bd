"""Host-frames mode against the pinned D2H ceiling of the box: 1280x800, 250-frame batches, 4 slots.
  ceiling : readback_async of resident frames only (no kernels), all slots' copy streams busy
  pipeline: submit + readback_async per slot (the D2H of batch i runs while the kernels of batch i + 1 do), with per-call wall times."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
dg = importlib.import_module("doom-rust-renderer_amd")
sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
path = np.fromfile(os.path.join(ROOT, "tests/golden/campath_seed1993.f32"), dtype="<f4").reshape(1000, 8)
sc = dg.Scene(sw.build_synth_iwad(1993), "e1m1")
W, H, B, S = 1280, 800, 250, 4
ctx = dg.Context(W, H, max_batch=B, slots=S); ctx.upload_scene(sc)
views = [dg.make_views(path[i * B:(i + 1) * B]) for i in range(4)]
bufs = [dg.lib().dg_alloc_host(B * ctx.frame_bytes) for _ in range(S)]
for s in range(S): ctx.submit(s, views[s])
for s in range(S): ctx.wait(s)
gb = B * ctx.frame_bytes / 1e9
for rep in range(2):
    t0 = time.perf_counter(); N = 16
    for i in range(N):
        if i >= S: ctx.wait(i % S)                      # one readback per slot at a time
        ctx.readback_async(i % S, 0, B, bufs[i % S])
    for s in range(S): ctx.wait(s)
    dt = time.perf_counter() - t0
    print(f"ceiling  (copies only)        : {N * gb / dt:6.1f} GB/s  = {N * B / dt:8.0f} frames/s", flush=True)
for rep in range(2):
    calls = []; t0 = time.perf_counter(); N = 16
    for i in range(N):
        a = time.perf_counter(); ctx.submit(i % S, views[i % S]); b = time.perf_counter(); ctx.readback_async(i % S, 0, B, bufs[i % S]); c = time.perf_counter()
        calls.append((b - a, c - b))
    for s in range(S): ctx.wait(s)
    dt = time.perf_counter() - t0
    t = ctx.timing(0)
    print(f"pipeline (render + copy)      : {N * gb / dt:6.1f} GB/s  = {N * B / dt:8.0f} frames/s | per batch {dt / N * 1e3:.2f} ms; submit call median "
          f"{np.median([c[0] for c in calls]) * 1e3:.2f} ms (it waits for the slot's previous copy), readback_async call {np.median([c[1] for c in calls]) * 1e3:.3f} ms; "
          f"kernels of one batch {t['total_ms']:.2f} ms", flush=True)
ref = ctx.readback(0, 0, 3)
got = np.ctypeslib.as_array((__import__('ctypes').c_uint8 * (3 * ctx.frame_bytes)).from_address(bufs[0])).reshape(3, H, W, 3)
print("first three frames of slot 0 in the pinned buffer equal dg_readback's:", bool(np.array_equal(ref, got)))
for b in bufs: dg.lib().dg_free_host(b)
ctx.close()
