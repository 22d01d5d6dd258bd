"""Which front end DG_FE_AUTO picks, batch by batch, in a pipelined run shaped like bench.py's side legs (manual GPU probe, not a test)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    args = bench.parse_args(["--config", str(cfg)])
    name, args.width, args.height, args.batch, maps, camera = bench.CONFIGS[cfg]
    be = bench.DoomGpuBackend(args, 0)
    ctx = be.load(maps[0], 1993, camera)
    n_slots, views = be.n_slots, be.views
    seq, ran = [], [False] * n_slots
    def step(g):
        s = g % n_slots
        if ran[s]:
            ctx.wait(s)
            t = ctx.timing(s)
            seq.append((t["front_end"], round(t["host_ms"], 3), round(t["setup_ms"], 3), round(t["raster_ms"], 3)))
        ctx.submit(s, views[s]); ran[s] = True
    step(0)
    for s in range(n_slots): ctx.wait(s)
    for g in range(1, 25): step(g)
    for s in range(n_slots): ctx.wait(s)
    print("front end per finished batch (2 = per-seg half on the host, 3 = seg walk), host ms, front-end kernel ms, raster ms:")
    for i, e in enumerate(seq): print(i, e)

main()
