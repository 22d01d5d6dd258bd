#!/usr/bin/env python3
"""bench.py — frames/sec of the MI355X rasteriser on the fixed e1m1 camera path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3,4,5}] [--width W --height H] [--batch 1000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Workload (config.workload): by default BASELINE config 3 — the 1 000-frame scripted camera path through the e1m1-like
map, rendered natively at 1280x800 (the size the ">= 10 000 fps" target is quoted on), one MI355X per rank.  No id
WAD can be shipped, so the maps are the committed synthetic IWADs and `data` says "synthetic" (with --wad FILE the same
command runs a real IWAD).  --config selects the other BASELINE.json configurations (CONFIGS below): 1 = the Player-1 start
frame at 320x200, 2 = the path at 320x200, 4 = eight paths over two maps at 1280x800, 5 = the heavy map at 2560x1600.

A *step* is ONE PASS OVER THE WHOLE 1 000-FRAME PATH: 1000 / `--batch` batches (default at 1280x800: one of 1000) of consecutive frames, each going through
the complete hot path of SURVEY.md §8(d) — host BSP walk / clip / projection / record generation on the ctx's host
threads, pinned staging, H2D, the device column walk (dg_fe_*) and the tile rasteriser dg_raster_tiles — leaving 1 000 RGB24
frames in HBM.  The batches are
pipelined over the ctx's slots: the host builds batch i + 1 while the GPU renders batch i.  `value` = frames / time over
exactly K steps.  This is what the reference's `Renderer::render()` (src/renderer/mod.rs:118-136) does per frame, with
`pixels.pixels` left in device memory; the rate with every frame also copied to host memory is `e2e_host_frames` (PCIe
bound, never `value`), the rate of the kernels alone on records already resident in HBM is `resident_replay`.

Multi-GPU: the path shards with no exchange step — rank r renders its own camera path (seed 1993 + r: the same closed
route entered at another room, odd seeds walk it backwards) and the ranks alternate between two maps (seeds 1993 / 1994,
the stand-ins for BASELINE config 4's map01 + map07) on GPU LOCAL_RANK; there is NO data-path collective and no RCCL.
torch.distributed (gloo, CPU tensors) is used only for the timing barrier, the MAX over ranks and gathering the per-rank
report; value = sum of frames / max time (weak scaling).  Each rank pins its host threads to its share of the CPUs
(the NUMA node of its GPU when the topology is readable) before anything touches HIP.

GPU clocks: before the W warm-up steps the same steps run untimed for --clock-warmup-ms (60): an MI355X that has been idle for 10 ms
renders its next launches 15-20 % slower and needs ~25 ms of work to be back at speed (tests/manual/gpu_warmup_probe.py); the line says
so (`gpu_clock_warmup`).  The K timed steps are exactly K, bracketed as the contract says.

roofline: the dominant kernel, dg_raster_tiles (one launch per batch), HBM-bound model.  achieved = algorithmic bytes per launch / mean duration from HIP events recorded on the kernel's own stream during the
timed steps (dg_slot_timing).  Algorithmic bytes per frame = 3*W*H (RGB24 stored
once) + W*H (one texel byte per pixel) + list bytes read (32 B per span, 4*(W+1) column index) — SURVEY.md §8d,
DESIGN.md "Roofline accounting".
cpu_baseline: the CPU oracle (oracle/doomref.c, a port of the reference renderer) on 1 host core over a bounded
sample of the same frames at the same size (rank 0, N = 1 only); the frames the timed steps left in HBM are compared with
it byte for byte.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PATH_FRAMES = 1000
# The synthetic stand-ins of BASELINE.json's maps: (generator seed, heavy).  Seed 1994 heavy = the 16x12-room map with more
# visplanes and sprites per frame (stand-in for config 4's second map and config 5's e2m1).
LIGHT_MAP, HEAVY_MAP = (1993, False), (1994, True)
# BASELINE.json configs -> (name, width, height, frames per batch, maps the ranks alternate between, camera: "path" | "start")
CONFIGS = {
    1: ("config 1: e1m1 stand-in at 320x200, the single Player-1 start viewpoint", 320, 200, 1000, (LIGHT_MAP,), "start"),
    2: ("config 2: e1m1 stand-in at 320x200, 1000-frame scripted camera path", 320, 200, 1000, (LIGHT_MAP,), "path"),
    3: ("config 3: e1m1 stand-in at 1280x800, 1000-frame scripted camera path", 1280, 800, 1000, (LIGHT_MAP,), "path"),
    4: ("config 4: two maps (map01 + map07 stand-ins) at 1280x800, eight independent camera paths, one per GPU", 1280, 800, 1000, (LIGHT_MAP, HEAVY_MAP), "path"),
    5: ("config 5: heavy map (e2m1 stand-in: more visplanes + sprites) at 2560x1600", 2560, 1600, 250, (HEAVY_MAP,), "path"),
}


def xorshift32(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= (x << 13) & 0xFFFFFFFF
    x ^= x >> 17
    x ^= (x << 5) & 0xFFFFFFFF
    return x & 0xFFFFFFFF


def seeded_route(route, path_seed: int):
    """Camera path `path_seed` (seeds 1993 ... 2000 = different start room / direction): the same closed route,
    entered at a seed-dependent waypoint; odd seeds above 1993 walk it backwards.  Seed 1993 is the route itself."""
    n = len(route)
    if path_seed == 1993 or n == 0:
        return list(route)
    k = xorshift32(path_seed * 2654435761) % n
    r = list(route[k:]) + list(route[:k])
    return r[::-1] if (path_seed - 1993) % 2 else r


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def rank_plan(rank: int, world: int, config: int = 3):
    """((map seed, heavy), path seed) of a rank: camera paths 1993 + rank, the config's maps alternating over the ranks."""
    maps = CONFIGS[config][4]
    return maps[rank % len(maps)], 1993 + rank


def pick_device(local_rank: int, visible: int, local_world: int = 0) -> int:
    """HIP ordinal of a rank: LOCAL_RANK; 0 when the launcher masked the GPUs per rank (HIP_VISIBLE_DEVICES = one GPU each, so every
    rank sees exactly one).  More local ranks than visible GPUs otherwise is an error: two ranks would share a GPU silently and the
    weak-scaling `value` would still be summed as if each had its own (DOOMGPU_BENCH_DEVICE overrides this to rehearse on one GPU)."""
    if visible <= 0:
        return local_rank                 # no GPU visible (CPU rehearsal): unchanged, dg_create reports it
    if visible == 1:
        return 0
    if local_rank >= visible or local_world > visible:
        raise SystemExit(f"bench.py: {max(local_world, local_rank + 1)} local ranks but only {visible} visible GPUs: one GPU per rank is the contract")
    return local_rank


def aggregate_fps(frames_per_rank: int, world: int, max_seconds: float) -> float:
    return frames_per_rank * world / max_seconds


def dist_max(seconds: float, dist) -> float:
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dist_all(ok: bool, dist) -> bool:
    """True when `ok` on every rank (the side measurements are collective: either every rank runs one or none does)."""
    if dist is None:
        return ok
    import torch
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def gather_reports(report: dict, dist, world: int):
    if dist is None:
        return [report]
    out = [None] * world
    dist.all_gather_object(out, report)
    return out


def _parse_cpulist(txt: str):
    cpus = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_nodes(sysfs: str = "/sys"):
    """NUMA node of every GPU in KFD enumeration order (= HIP ordinal order on a default box), read from sysfs without
    touching HIP; [] when the topology cannot be read.  (`sysfs`: the tree to read — tests point it at a made-up 8-GPU box.)"""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    nodes = []
    try:
        for d in sorted(os.listdir(base), key=int):
            props = dict(l.split() for l in open(os.path.join(base, d, "properties")) if len(l.split()) == 2)
            if int(props.get("simd_count", "0")) == 0:
                continue
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
            try:
                nodes.append(int(open(os.path.join(sysfs, f"bus/pci/devices/{bdf}/numa_node")).read()))
            except OSError:
                nodes.append(-1)
    except (OSError, ValueError):
        return []
    return nodes


def rank_cpu_share(local_rank: int, local_world: int, allowed, sysfs: str = "/sys"):
    """-> (CPUs of this rank, how they were chosen): the CPUs of the GPU's NUMA node split among the ranks whose GPUs sit on that
    node, else an even split of the allowed CPUs."""
    allowed = sorted(allowed)
    if local_world <= 1 or len(allowed) < local_world:
        return allowed, "all allowed CPUs"
    share, how = None, ""
    numa = gpu_numa_nodes(sysfs)
    if len(numa) >= local_world and numa[local_rank] >= 0:
        node = numa[local_rank]
        try:
            node_cpus = [c for c in _parse_cpulist(open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read()) if c in set(allowed)]
            peers = [r for r in range(local_world) if numa[r] == node]
            if len(node_cpus) >= len(peers):
                i, n = peers.index(local_rank), len(peers)
                share = node_cpus[i * len(node_cpus) // n:(i + 1) * len(node_cpus) // n]
                how = f"NUMA node {node} shared by {n} rank(s)"
        except OSError:
            share = None
    if not share:
        share = allowed[local_rank * len(allowed) // local_world:(local_rank + 1) * len(allowed) // local_world]
        how = "even split of the allowed CPUs"
    return share, how


def bind_rank_cpus(local_rank: int, local_world: int):
    """Pin this process (and the threads it will create) to its share of the host CPUs (rank_cpu_share) BEFORE any HIP call."""
    share, how = rank_cpu_share(local_rank, local_world, os.sched_getaffinity(0))
    if local_world > 1 and len(share) >= 1:
        os.sched_setaffinity(0, share)
    return share, how


def cgroup_cpu_quota(root: str = "/sys/fs/cgroup") -> int:
    """CPUs' worth of run time the container allows this process (cgroup v2 cpu.max, v1 cfs quota), rounded up; 0 = no limit / unknown."""
    try:
        q, p = open(os.path.join(root, "cpu.max")).read().split()[:2]
        return 0 if q == "max" else -(-int(q) // int(p))
    except (OSError, ValueError):
        pass
    try:
        q = int(open(os.path.join(root, "cpu/cpu.cfs_quota_us")).read())
        p = int(open(os.path.join(root, "cpu/cpu.cfs_period_us")).read())
        return -(-q // p) if q > 0 and p > 0 else 0
    except (OSError, ValueError):
        return 0


def default_host_threads(local_world: int = 1, cgroup_root: str = "/sys/fs/cgroup") -> int:
    """Host threads of this rank's ctx: its CPU share — the affinity mask bind_rank_cpus has set by now and this rank's part of the
    container's CPU quota (the GPU boxes show 256 CPUs and allow 16 CPUs of run time per GPU: 64 threads there get the process
    throttled, 2560x1600 drops from 133 k to 102 k frames/s) — at most 16, the count every number in profiles/ was measured with (more
    do not help the condition-variable pool: 0.95 / 0.85-1.1 ms per 1 000-frame batch with 16 / 32 threads)."""
    n = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota(cgroup_root)
    if quota > 0:
        n = min(n, max(1, quota // max(1, local_world)))
    return max(1, min(n, 16))


class DoomGpuBackend:
    """The real thing: libdoomgpu through the ctypes binding, synthetic IWADs, the camera-path generator."""

    def __init__(self, args, device: int):
        self.dg = importlib.import_module("doom-rust-renderer_amd")
        self.sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
        self.cp = importlib.import_module("doom-rust-renderer_amd.camera_path")
        self.args, self.device = args, device

    def load(self, map_id, path_seed: int, camera: str = "path"):
        import numpy as np
        a = self.args
        map_seed, heavy = map_id
        if a.wad:
            self.wad = open(a.wad, "rb").read()          # a user-supplied IWAD (BASELINE configs verbatim); data = "file"
            route = self.cp.route_from_wad(self.wad, a.map)
        elif getattr(a, "synth_map", None):               # SEED:COLUMNSxROWS:THINGS, heavy + vanilla (2002:32x24:500 = the doom2-scale map of the GPU tier)
            seed, grid, things = a.synth_map.split(":")
            kw = dict(heavy=True, vanilla=True, grid=tuple(int(v) for v in grid.split("x")), n_things=int(things))
            self.wad = self.sw.build_synth_iwad(int(seed), **kw)
            route = self.sw.synth_route(int(seed), **kw)
        else:
            self.wad = self.sw.build_synth_iwad(map_seed, heavy=heavy)
            route = self.sw.synth_route(map_seed, heavy=heavy)
        self.scene = self.dg.Scene(self.wad, a.map)
        if camera == "start":                             # BASELINE config 1: the Player-1 start, Game::new's view (src/game.rs:151-156)
            x, y, ang = self.scene.player_start()
            self.path = np.tile(self.cp.view_record(x, y, ang, self.scene.floor_height_at(x, y, 0.0)), (PATH_FRAMES, 1))
        else:
            self.path = self.cp.make_camera_path(seeded_route(route, path_seed), lambda x, y, d: self.scene.floor_height_at(x, y, d), PATH_FRAMES)
        B = a.batch
        self.n_slots = max(1, a.slots)                # (with fewer batches per step than slots, a slot holds the same batch every other step)
        fe = {"host": self.dg.DG_FE_HOST, "device": self.dg.DG_FE_DEVICE, "segs": self.dg.DG_FE_DEVICE_SEGS, "auto": self.dg.DG_FE_AUTO}[a.front_end]
        self.ctx = self.dg.Context(a.width, a.height, max_batch=B, slots=self.n_slots, device=self.device, host_threads=a.host_threads or default_host_threads(getattr(a, "local_world", 1)), front_end=fe)
        self.ctx.upload_scene(self.scene)
        loop = np.concatenate([self.path, self.path])
        self.batch_first = [s * B % PATH_FRAMES for s in range(self.n_slots)]
        self.views = [self.dg.make_views(loop[b0:b0 + B]) for b0 in self.batch_first]
        return self.ctx

    def oracle(self):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import doomref
        return doomref


def run(args, backend_factory=DoomGpuBackend):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpus, cpu_how = bind_rank_cpus(local_rank, local_world)           # before torch / HIP are touched
    args.local_world = local_world                                    # (default_host_threads: this rank's part of a container CPU quota)

    # torch first: its bundled HIP runtime must be the one libdoomgpu.so binds to (same soname, loaded once).
    import torch
    visible = torch.cuda.device_count()                               # counting devices does not initialise HIP on this image
    device = int(os.environ["DOOMGPU_BENCH_DEVICE"]) if "DOOMGPU_BENCH_DEVICE" in os.environ else pick_device(local_rank, visible, local_world)   # override only to rehearse N > 1 on a 1-GPU box
    print(f"[bench rank {rank}/{world}] local_rank {local_rank}, {visible} visible GPU(s) -> device {device}; {len(cpus)} CPUs ({cpu_how})", file=sys.stderr, flush=True)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # gloo prints its "[Gloo] Rank r is connected to ..." banner on fd 1; stdout must carry only the JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo")     # timing barrier / MAX / report gather only; the data path has no collective
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    if torch.cuda.is_available():
        torch.cuda.set_device(device)

    import numpy as np
    W, H, B = args.width, args.height, args.batch
    map_id, path_seed = rank_plan(rank, world, args.config)
    map_seed = map_id[0]
    be = backend_factory(args, device)
    ctx = be.load(map_id, path_seed, CONFIGS[args.config][5])
    n_slots, views = be.n_slots, be.views
    batches_per_step = PATH_FRAMES // B                # --batch divides the path (parse_args)
    frames_per_step = batches_per_step * B

    def sync_all():
        for s in range(n_slots):
            ctx.wait(s)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    submitted = [0]

    def one_pass(collect=None):
        """One step: the whole path, batch by batch, through the complete hot path (slot g % n_slots holds batch g of the endless
        walk round the path: with fewer batches per step than slots the pipeline is simply deeper than one step)."""
        for _ in range(batches_per_step):
            s = submitted[0] % n_slots
            submitted[0] += 1
            if collect is not None and ctx_ran[s]:
                ctx.wait(s)
                collect(s)
            ctx.submit(s, views[s])               # host record generation (blocking) + H2D + kernels (queued); waits only if the slot is still busy
            ctx_ran[s] = True

    # ---- the headline: warmup, then exactly K timed steps ------------------------------------------------------------
    ctx_ran = [False] * n_slots
    # The GPU's clocks first: an MI355X that has been idle for even 10 ms renders its next launches 15-20 % slower and takes ~25 ms of
    # continuous work to be back at speed (tests/manual/gpu_warmup_probe.py: dg_raster_tiles 1.88, 1.97, 1.85, ... 1.62 ms over 14 launches,
    # after a cold start and after every pause alike) — W = 2 warm-up steps are 4 ms.  So, before the W warm-up steps of the contract, the
    # same steps run untimed until --clock-warmup-ms have passed (reported as `gpu_clock_warmup`); the K timed steps then measure the
    # renderer, not the power management's ramp.
    clock_warm_steps, t_warm, warm_raster_ms = 0, time.perf_counter(), []
    while (time.perf_counter() - t_warm) * 1e3 < args.clock_warmup_ms:
        one_pass(lambda slot: warm_raster_ms.append(round(ctx.timing(slot)["raster_ms"], 3)))     # (the ramp itself goes into the line: the evidence)
        clock_warm_steps += 1
    for _ in range(args.warmup):
        one_pass()
    sync_all()
    barrier()
    sync_all()
    raster_ms, setup_ms, host_ms, alg_bytes = [], [], [], []
    stats = {}
    fe_used = {}

    def collect(slot):
        t = ctx.timing(slot)
        raster_ms.append(t["raster_ms"])
        setup_ms.append(t["setup_ms"])
        host_ms.append(t["host_ms"])
        nf = t["n_frames"]
        alg_bytes.append(nf * (3 * W * H + W * H + 4 * (W + 1)) + 32 * t["n_spans"])
        fe_used[t.get("front_end", 2)] = fe_used.get(t.get("front_end", 2), 0) + 1
        if t.get("front_end", 2) != 3 or "host_ms" not in stats:
            stats.update(t)

    t0 = time.perf_counter()
    step_end = []
    for _ in range(args.steps):
        one_pass(collect)
        step_end.append(time.perf_counter())          # (a submission blocks while its slot is busy: in steady state this is the GPU's pace)
    sync_all()
    elapsed_local = time.perf_counter() - t0
    step_end[-1] = t0 + elapsed_local                 # the last step ends when the pipeline has drained
    step_ms = [1e3 * (b - a) for a, b in zip([t0] + step_end[:-1], step_end)]
    # the first steps only fill the slots (a submission returns as soon as its host work is done) and the last one ends with the drain:
    # the spread is taken over the steps in between, where a submission waits for the GPU to free its slot
    steady = step_ms[n_slots // batches_per_step + 1:-1] if len(step_ms) > n_slots // batches_per_step + 4 else step_ms
    barrier()
    sync_all()
    for s in range(n_slots):
        if ctx_ran[s]:
            collect(s)
    elapsed = dist_max(elapsed_local, dist)
    value = aggregate_fps(args.steps * frames_per_step, world, elapsed)
    fallbacks = ctx.fallbacks() if hasattr(ctx, "fallbacks") else None

    # ---- CPU baseline + parity of the frames the timed steps left in HBM (rank 0, N = 1 only) -----------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, be, ctx, n_slots, np)

    # ---- the kernels alone: records resident in HBM, replayed (no host work, no H2D) --------------------------------
    resident = None
    iso_ms = []
    if not args.no_resident:
        for s in range(n_slots):
            ctx.prepare(s, views[s])
        for i in range(n_slots):
            ctx.replay(i)
        sync_all()
        barrier()
        t1 = time.perf_counter()
        nrep = max(2 * n_slots, min(args.steps * batches_per_step, 80))
        for i in range(nrep):
            s = i % n_slots
            if i >= n_slots:
                ctx.wait(s)
            ctx.replay(s)
        sync_all()
        rs = dist_max(time.perf_counter() - t1, dist)
        resident = {"value": aggregate_fps(nrep * B, world, rs), "unit": "frames/s",
                    "includes": "device column walk + rasteriser on per-seg / per-sprite records already resident in HBM"}
        # the rasteriser with nothing else on the GPU: replays with a wait in between
        for i in range(2 * n_slots):
            ctx.replay(i % n_slots)
            ctx.wait(i % n_slots)
            t = ctx.timing(i % n_slots)
            iso_ms.append(t["raster_ms"])
        iso_ms = iso_ms[n_slots:]

    # ---- every frame also copied to page-locked host memory (what the reference's `pixels.pixels` literally is) -------
    e2e_host = None
    if not args.no_host_frames and hasattr(ctx, "readback_async"):
        e2e_host = host_frames_rate(args, be, ctx, n_slots, views, B, batches_per_step, barrier, dist, world)

    # ---- the drop-in call shape: ONE view per call, synchronously, into caller memory (rank 0, N = 1 only) ------------
    latency = None
    if rank == 0 and world == 1 and not args.no_latency and hasattr(ctx, "render_one_into"):
        try:
            latency = single_frame_latency(args, be, device, np)
        except Exception as e:                            # a side measurement must not take the headline line down with it
            latency = {"error": repr(e)}

    # ---- the other configurations through the same timed path (rank 0, N = 1 only) --------------------------------------
    side = None
    if rank == 0 and world == 1 and not args.no_side_legs and backend_factory is DoomGpuBackend and args.config == 3 and not args.wad and not args.synth_map:
        side = {}
        # (a config-2 step is 0.35 ms: 40 of them; config 2 a second time with the device seg walk forced: what the kernels do when DG_FE_AUTO does
        # not spend one of the leg's batches timing the host walker — it does that once per context, 1.2 ms against 0.3)
        for key, (cfg, ht, k, fe) in {"config2": (2, 0, 40, None), "config2_seg_walk": (2, 0, 40, "segs"), "config5": (5, 0, 5, None),
                                      "config3_two_host_threads": (3, 2, 5, None)}.items():
            try:
                side[key] = side_leg(args, backend_factory, device, np, cfg, k, ht, fe)
            except Exception as e:                        # a side measurement must not take the headline line down with it
                side[key] = {"error": repr(e)}

    # ---- roofline of the rasteriser --------------------------------------------------------------------------------
    mean_raster_s = float(np.mean(raster_ms)) / 1e3
    achieved = float(np.mean(alg_bytes)) / mean_raster_s / 1e9
    traffic, achievable_fill = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(f"{W}x{H}x{B}", {}).get("hbm_bytes_per_launch")
            achievable_fill = tj.get("_achievable", {}).get("fill_GBps")      # measured on the box with tools/microbench/hbm_copy.py
        except Exception:
            traffic = None
    roofline = {"kernel": "dg_raster_tiles",
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                "algorithmic_bytes_per_launch": float(np.mean(alg_bytes)), "mean_launch_ms": mean_raster_s * 1e3,
                "front_end_kernels_mean_ms": float(np.mean(setup_ms)),
                "frames_per_launch": B, "pixels_per_s": B * W * H / mean_raster_s,
                "achievable_write_GBps": achievable_fill, "frac_of_achievable_write": (achieved / achievable_fill) if achievable_fill else None,
                "note": "achieved/frac are measured over the timed steps (all kernels on one in-order stream; only the next batch's H2D copy overlaps these kernels); "
                        "isolated_* is the same launch measured with nothing else on the GPU"}
    roofline["issue"] = issue_roofline(W, H, B, mean_raster_s)
    if iso_ms:
        iso = float(np.mean(iso_ms)) / 1e3
        roofline.update({"isolated_launch_ms": iso * 1e3,
                         "isolated_achieved": float(np.mean(alg_bytes)) / iso / 1e9, "isolated_frac": float(np.mean(alg_bytes)) / iso / 1e9 / 8000.0})

    report = {"rank": rank, "map_seed": map_seed, "heavy_map": bool(map_id[1]), "path_seed": path_seed, "device": device, "frames_per_s": args.steps * frames_per_step / elapsed_local,
              "host_ms_per_batch": float(np.mean(host_ms)), "host_threads": getattr(ctx, "host_threads", None), "cpus": len(cpus), "cpu_binding": cpu_how}
    reports = gather_reports(report, dist, world)

    line = None
    if rank == 0:
        names = {1: "host span lists", 2: "device column walk (per-seg half on the host)", 3: "device seg walk + device column walk (nothing on the host)"}
        fe_name = "; ".join(f"{names[k]}: {v} batch(es)" for k, v in sorted(fe_used.items())) or names[2]
        line = {
            "metric": "frames/sec (fixed e1m1 camera path)", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            # spread over the timed steps (rank 0's host clock after each step's submissions), taken over the steady ones: the first steps
            # fill the slots, the last ends with the drain (both given separately); value / ms_per_step above are the contract's K-step mean
            "ms_per_step_median": float(np.median(steady)), "ms_per_step_min": float(np.min(steady)), "ms_per_step_max": float(np.max(steady)),
            "ms_per_step_first": float(step_ms[0]), "ms_per_step_last": float(step_ms[-1]),
            "gpu_ms_per_batch": {"median": float(np.median(np.add(raster_ms, setup_ms))), "min": float(np.min(np.add(raster_ms, setup_ms))),
                                 "max": float(np.max(np.add(raster_ms, setup_ms))), "what": "front-end kernels + raster launch of each timed batch (HIP events on the kernel stream)"},
            "gpu_clock_warmup": {"ms": args.clock_warmup_ms, "steps": clock_warm_steps, "raster_ms_of_its_launches": warm_raster_ms[:24],
                                 "why": "untimed steps before the warm-up steps: after any idle the GPU needs ~25 ms of work to reach its clocks (tests/manual/gpu_warmup_probe.py)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "file" if args.wad else "synthetic",
            "config": {"workload": (f"NOT a BASELINE configuration: generated map {args.synth_map} (seed:rooms:things, heavy + vanilla) at the size of " if args.synth_map and not args.wad else "BASELINE ") +
                                   f"{CONFIGS[args.config][0]}; " +
                                   (f"{os.path.basename(args.wad)} {args.map}" if args.wad else f"synthetic IWAD {args.synth_map}" if args.synth_map else
                                    "synthetic IWAD(s) " + " + ".join(f"seed {m}{' (heavy)' if hv else ''}" for (m, hv) in CONFIGS[args.config][4]) + " (no id WAD in the environment)") +
                                   f", {W}x{H} native; one step = 1000 frames in {batches_per_step} batches of {B} through "
                                   "host record generation + H2D + all kernels, frames left in HBM (SURVEY 8d)",
                       "baseline_config": args.config,
                       "front_end": fe_name, "width": W, "height": H, "frames_per_step": frames_per_step, "frames_per_batch": B, "slots": n_slots,
                       "parallelism": f"{world} independent camera path(s) (seeds 1993..{1993 + world - 1}), one per GPU, maps alternating over the ranks "
                                      f"{[m for (m, _) in CONFIGS[args.config][4]]}, no collective"},
            "roofline": roofline, "cpu_baseline": cpu, "resident_replay": resident, "e2e_host_frames": e2e_host, "latency": latency,
            "host": {"ms_per_batch": float(np.mean(host_ms)), "threads": getattr(ctx, "host_threads", None),
                     "list_bytes_per_frame": int(stats.get("list_bytes", 0) // max(1, stats.get("n_frames", 1)))},
            "fallbacks": fallbacks, "side_legs": side, "per_rank": reports,
        }
        print(json.dumps(line))
    ctx.close()
    if dist is not None:
        dist.barrier()
    return line


def cpu_baseline(args, be, ctx, n_slots, np):
    W, H, B = args.width, args.height, args.batch
    doomref = be.oracle()
    path = be.path
    osc = doomref.Scene(be.wad, args.map)
    idx = list(range(0, PATH_FRAMES, max(1, args.cpu_sample)))
    refs = []
    tc = time.perf_counter()
    for i in idx:
        refs.append(osc.render(W, H, path[i]))
    dt = time.perf_counter() - tc
    cpu = {"value": len(idx) / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"every {args.cpu_sample}th frame of the same 1000-frame path at {W}x{H} ({len(idx)} frames, {dt:.1f} s), "
                     "oracle/doomref.c -O2 -ffp-contract=off, cos/sin hoisted per frame",
           "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    # the same oracle with the sampled frames sharded over the host cores this process may use (one scene per thread: the
    # oracle's lazy texture caches are per scene; ctypes releases the GIL around dr_render)
    try:
        import threading
        ncores = max(1, min(len(os.sched_getaffinity(0)), 64))
        scenes = [doomref.Scene(be.wad, args.map) for _ in range(ncores)]
        bufs = [np.empty(3 * W * H, dtype=np.uint8) for _ in range(ncores)]
        for sc_t, bt in zip(scenes, bufs):
            sc_t.render(W, H, path[0], out=bt.ctypes.data)    # warm the lazy caches outside the clock

        def work(t):
            for i in idx[t::ncores]:
                scenes[t].render(W, H, path[i], out=bufs[t].ctypes.data)
        th = [threading.Thread(target=work, args=(t,)) for t in range(ncores)]
        tm = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dtm = time.perf_counter() - tm
        cpu["all_cores"] = {"value": len(idx) / dtm, "unit": "frames/s", "cores": ncores,
                            "sample": f"the same {len(idx)} frames sharded over {ncores} threads ({dtm:.2f} s)"}
    except Exception as e:                            # the single-core figure above is the contract; this one is extra
        cpu["all_cores"] = {"error": str(e)}
    # parity of the frames the timed, pipelined steps left in the slots: every sampled frame, byte for byte
    bad = checked = 0
    distinct = set()
    for s in range(n_slots):
        ctx.wait(s)
        got = ctx.readback(s, 0, B)
        first = be.batch_first[s]
        for k, i in enumerate(idx):
            j = (i - first) % PATH_FRAMES
            if j < B:
                ref = np.frombuffer(refs[k], dtype=np.uint8).reshape(H, W, 3)
                bad += not np.array_equal(got[j], ref)
                checked += 1
                distinct.add(i)
    cpu["gpu_frames_checked"] = checked               # slot frames compared (with more slots than batches per step a batch sits in two slots)
    cpu["gpu_distinct_frames_checked"] = len(distinct)
    cpu["gpu_frames_bit_exact"] = bool(bad == 0)
    cpu["gpu_frames_source"] = "the slots as the timed pipelined steps left them"
    return cpu


def single_frame_latency(args, be, device, np):
    """What the reference's game loop does per displayed frame (src/game.rs:505-525): `Pixels::new()`, `Renderer::new(..).render()`, then
    `pixels.pixels` is in host memory — here dg_render_views(ctx, &view, 1, host_ptr) on a max_batch = 1 context (rust/src/gpu.rs
    GpuRenderer::render), one call per frame of the 1 000-frame path, at the reference's native 1024x768 (src/game.rs:28-29) and at the
    headline size.  Wall-clock per call: host records for one view, H2D, all kernels, D2H of the RGB24 frame, every wait.  The oracle's
    time per frame (one core) on a sample of the same views stands beside it.  Never `value`."""
    dg = be.dg
    doomref = be.oracle()
    osc = doomref.Scene(be.wad, args.map)
    out = {"call": "dg_render_views(ctx, &view, 1, host_ptr), max_batch 1, synchronous", "frames": PATH_FRAMES, "sizes": {}}
    for (W, H) in ((1024, 768), (args.width, args.height)) if (args.width, args.height) != (1024, 768) else ((1024, 768),):
        ctx1 = dg.Context(W, H, max_batch=1, slots=1, device=device, host_threads=1)
        try:
            ctx1.upload_scene(be.scene)
            views = dg.make_views(be.path)
            nbytes = 3 * W * H
            res = {}
            pinned = dg.lib().dg_alloc_host(nbytes)
            pageable = np.empty(nbytes, dtype=np.uint8)
            try:
                for name, ptr in (("pinned", pinned), ("pageable", pageable.ctypes.data)):
                    if not ptr:
                        continue
                    for i in range(8):                                    # untimed: first touches of the target and of the ctx's paths
                        ctx1.render_one_into(views[i], ptr)
                    dts = np.empty(PATH_FRAMES)
                    for i in range(PATH_FRAMES):
                        ta = time.perf_counter()
                        ctx1.render_one_into(views[i], ptr)
                        dts[i] = time.perf_counter() - ta
                    res[name] = {"median_ms": float(np.median(dts) * 1e3), "p99_ms": float(np.percentile(dts, 99) * 1e3), "mean_ms": float(dts.mean() * 1e3),
                                 "frames_per_s": float(PATH_FRAMES / dts.sum())}
                # the last frame of the pageable run against the oracle, byte for byte
                ref = np.frombuffer(osc.render(W, H, be.path[PATH_FRAMES - 1]), dtype=np.uint8)
                res["last_frame_bit_exact"] = bool(np.array_equal(pageable, ref))
            finally:
                if pinned:
                    dg.lib().dg_free_host(pinned)
            idx = list(range(0, PATH_FRAMES, 50))
            tc = time.perf_counter()
            for i in idx:
                osc.render(W, H, be.path[i])
            res["cpu_oracle_ms_per_frame"] = (time.perf_counter() - tc) / len(idx) * 1e3
            res["cpu_oracle_sample"] = f"every 50th frame of the same path ({len(idx)} frames), oracle/doomref.c on 1 core"
            out["sizes"][f"{W}x{H}"] = res
        finally:
            ctx1.close()
    return out


def host_frames_rate(args, be, ctx, n_slots, views, B, batches_per_step, barrier, dist, world):
    """Every frame to page-locked host memory, D2H of batch i overlapped with the kernels of batch i + 1."""
    dg = be.dg
    n_slots = min(n_slots, 2)                         # two slots are enough to overlap copy and kernels; each needs B frames of page-locked memory
    nb = max(2 * n_slots, min(4 * batches_per_step, 16))
    bufs = [dg.lib().dg_alloc_host(B * ctx.frame_bytes) for _ in range(n_slots)]
    try:
        if not dist_all(all(bufs), dist):             # a rank that could not page-lock its buffers: nobody measures this mode
            return None
        for s in range(n_slots):
            ctx.wait(s)
        for s in range(n_slots):                      # untimed: the first copy into a fresh page-locked buffer pays its page faults
            ctx.readback_async(s, 0, B, bufs[s])
        for s in range(n_slots):
            ctx.wait(s)
        barrier()
        t2 = time.perf_counter()
        err = None
        try:
            for i in range(nb):
                s = i % n_slots
                ctx.submit(s, views[s])               # waits for the slot's previous readback if it is still in flight
                ctx.readback_async(s, 0, B, bufs[s])  # queued behind the slot's kernels on its copy stream
            for s in range(n_slots):
                ctx.wait(s)
        except Exception as e:                        # a side measurement must not take the headline line down with it
            err = repr(e)
        h_s = dist_max(time.perf_counter() - t2, dist)
        if not dist_all(err is None, dist):
            return {"error": err or "failed on another rank"}
    finally:
        for b in bufs:
            if b:
                dg.lib().dg_free_host(b)
    return {"value": aggregate_fps(nb * B, world, h_s), "unit": "frames/s",
            "includes": "the headline path + D2H of every RGB24 frame to pinned host memory, copy of batch i overlapped with the kernels of batch i + 1",
            "d2h_GBps": nb * B * ctx.frame_bytes / h_s / 1e9}


def side_leg(args, backend_factory, device, np, config: int, steps: int, host_threads: int = 0, front_end=None):
    """One more BASELINE configuration through the WHOLE timed path of the headline (host record generation + H2D + every kernel, frames
    left in HBM): a context of its own, one warm-up step, `steps` timed steps bracketed by waits.  Never `value`: these are the numbers the
    other configurations (and the headline with two host threads) run at, in the driver's record instead of a builder's log."""
    import copy
    a = copy.copy(args)
    name, a.width, a.height, a.batch, maps, camera = CONFIGS[config]
    a.config, a.host_threads, a.wad, a.synth_map = config, host_threads, None, None
    if front_end:
        a.front_end = front_end
    W, H, B = a.width, a.height, a.batch
    be = backend_factory(a, device)
    ctx = be.load(maps[0], 1993, camera)
    n_slots, views = be.n_slots, be.views
    batches_per_step = PATH_FRAMES // B
    ran = [False] * n_slots
    raster_ms, setup_ms, host_ms, alg, fe_used = [], [], [], [], {}

    def collect(s):
        t = ctx.timing(s)
        raster_ms.append(t["raster_ms"]); setup_ms.append(t["setup_ms"]); host_ms.append(t["host_ms"])
        alg.append(t["n_frames"] * (3 * W * H + W * H + 4 * (W + 1)) + 32 * t["n_spans"])
        fe_used[t.get("front_end", 2)] = fe_used.get(t.get("front_end", 2), 0) + 1

    def one_pass(g0, timed):
        for g in range(g0, g0 + batches_per_step):
            s = g % n_slots
            if ran[s]:
                ctx.wait(s)
                if timed:
                    collect(s)
            ctx.submit(s, views[s])
            ran[s] = True
        return g0 + batches_per_step

    g = one_pass(0, False)
    t_warm = time.perf_counter()                      # (the GPU's clocks, as in run(): untimed steps for --clock-warmup-ms)
    while (time.perf_counter() - t_warm) * 1e3 < getattr(args, "clock_warmup_ms", 0.0):
        g = one_pass(g, False)
    for s in range(n_slots):
        ctx.wait(s)
    t0 = time.perf_counter()
    for _ in range(steps):
        g = one_pass(g, True)
    for s in range(n_slots):
        ctx.wait(s)
    elapsed = time.perf_counter() - t0
    for s in range(n_slots):
        if ran[s]:
            collect(s)
    fb = ctx.fallbacks() if hasattr(ctx, "fallbacks") else None
    ctx.close()
    names = {1: "host span lists", 2: "device column walk", 3: "device seg walk + column walk"}
    mean_r = float(np.mean(raster_ms)) / 1e3
    return {"workload": f"BASELINE {name}, {W}x{H}, {batches_per_step} batch(es) of {B} per step" + (f", {host_threads} host threads" if host_threads else "") + (f", --front-end {front_end}" if front_end else ""),
            "baseline_config": config, "steps": steps, "warmup": 1, "value": steps * batches_per_step * B / elapsed, "unit": "frames/s", "ms_per_step": elapsed / steps * 1e3,
            "raster_ms_per_batch": mean_r * 1e3, "front_end_kernels_ms_per_batch": float(np.mean(setup_ms)), "host_ms_per_batch": float(np.mean(host_ms)),
            "host_threads": getattr(ctx, "host_threads", None) or host_threads or None,
            "roofline_frac": float(np.mean(alg)) / mean_r / 1e9 / 8000.0,
            "front_end": "; ".join(f"{names.get(k, k)}: {v} batch(es)" for k, v in sorted(fe_used.items())), "fallbacks": fb}


def issue_roofline(W, H, B, mean_launch_s):
    """`roofline.issue`: dg_raster_tiles is bound by instruction issue, not by bandwidth, so the line also says how many
    wave-instructions of each class one (column, 64-row) chunk costs and what share of the issue capacity they fill at the launch time THIS
    run measured.  The counts cannot be read from inside the process: they come from profiles/issue.json, written by
    tools/issue_counters.py from rocprofv3 --pmc passes of the same build and launch shape (SQ_INSTS_*, SQ_ACTIVE_INST_VALU2 = vector
    instructions that shared an issue slot, SQ_LDS_IDX_ACTIVE, SQ_BUSY_CU_CYCLES for the clock).  Capacities (profiles/r05_issue_model.md,
    measured with tools/microbench/issue_rates.hip under the same counters): one vector issue slot per SIMD every 4 clocks, into which a
    second instruction of the simple class can be paired; one scalar instruction per CU and clock; one LDS cycle per CU and clock."""
    path = os.path.join(ROOT, "profiles", "issue.json")
    if not os.path.exists(path):
        return None
    try:
        j = json.load(open(path)).get(f"{W}x{H}x{B}")
        if not j:
            return None
        chunks = B * ((W + 63) // 64) * 64 * ((H + 63) // 64)            # (column, 64-row) chunks per launch
        clocks = mean_launch_s * j["clock_ghz"] * 1e9
        cus, simds = 256, 1024
        valu_slots = j["valu"] - j["valu_paired"]
        return {"per_chunk": {k: j[k] / chunks for k in ("valu", "valu_paired", "salu", "lds", "vmem", "smem")},
                "per_chunk_total": sum(j[k] for k in ("valu", "salu", "lds", "vmem", "smem")) / chunks,
                "vector_issue_slots_filled": valu_slots * 4.0 / (clocks * simds),
                "scalar_issue_filled": j["salu"] / (clocks * cus),
                "lds_cycles_filled": j["lds_cycles"] / (clocks * cus),
                "clock_ghz": j["clock_ghz"], "chunks_per_launch": chunks,
                "source": "profiles/issue.json (rocprofv3 --pmc of this build, tools/issue_counters.py); fractions use this run's mean launch time"}
    except Exception as e:
        return {"error": repr(e)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clock-warmup-ms", type=float, default=60.0, help="untimed steps run for this long before the --warmup steps, so that the GPU's clocks "
                                                                         "are up when the timed steps start (0: none)")
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=3, help="BASELINE.json configuration (1-based); sets width / height / batch / maps")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="frames per batch; must divide the 1000-frame path")
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--front-end", choices=["auto", "device", "host", "segs"], default="auto",
                    help="host: everything of the front end on the host; device: the per-column half on the GPU (DG_FE_DEVICE); segs: the per-seg half too "
                         "(DG_FE_DEVICE_SEGS); auto (default, DG_FE_AUTO): device or segs per batch, whichever the library measures to be faster")
    ap.add_argument("--cpu-sample", type=int, default=4, help="cpu_baseline renders every n-th path frame")
    ap.add_argument("--wad", default=None, help="IWAD file to render instead of the synthetic one (e.g. doom1.wad); the camera path is derived from the map")
    ap.add_argument("--map", default="e1m1")
    ap.add_argument("--synth-map", default=None, help="SEED:COLUMNSxROWS:THINGS: a generated vanilla-shaped map of that many rooms and things instead of the config's "
                                                     "(2002:32x24:500 = 18 131 segs, doom2's scale); not a BASELINE configuration, `config.workload` says so")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resident", action="store_true")
    ap.add_argument("--no-host-frames", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-view-per-call latency leg (`latency` in the JSON line)")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the short legs of configs 2 and 5 and of config 3 with two host threads (`side_legs` in the JSON line)")
    ap.add_argument("--latency", action="store_true", help="(default) measure the one-view-per-call latency leg at 1024x768 and the bench size")
    a = ap.parse_args(argv)
    _, cw, ch, cb, _, _ = CONFIGS[a.config]
    a.width = a.width or cw
    a.height = a.height or ch
    a.batch = a.batch or cb
    if a.batch <= 0 or PATH_FRAMES % a.batch:
        ap.error(f"--batch must divide the {PATH_FRAMES}-frame path (a step is one pass over the whole path)")
    return a


def main(argv=None, backend_factory=DoomGpuBackend):
    args = parse_args(argv)
    import torch.distributed as dist
    try:
        return run(args, backend_factory)
    finally:
        if dist.is_available() and dist.is_initialized() and backend_factory is DoomGpuBackend:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
