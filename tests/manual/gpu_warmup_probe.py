"""How long dg_raster_tiles takes launch by launch from a cold start, and again after a pause (manual GPU probe: clock ramp or first touch?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

def main():
    args = bench.parse_args(["--config", "3"])
    name, args.width, args.height, args.batch, maps, camera = bench.CONFIGS[3]
    be = bench.DoomGpuBackend(args, 0)
    ctx = be.load(maps[0], 1993, camera)
    n_slots, views = be.n_slots, be.views
    def burst(n, tag):
        out, ran = [], [False] * n_slots
        for g in range(n):
            s = g % n_slots
            if ran[s]:
                ctx.wait(s); out.append(round(ctx.timing(s)["raster_ms"], 3))
            ctx.submit(s, views[s]); ran[s] = True
        for s in range(n_slots):
            if ran[s]:
                ctx.wait(s); out.append(round(ctx.timing(s)["raster_ms"], 3))
        print(tag, out)
    burst(28, "cold start:")
    burst(12, "straight on:")
    for pause in (0.01, 0.1, 1.0):
        time.sleep(pause)
        burst(16, f"after a pause of {pause} s:")

main()
