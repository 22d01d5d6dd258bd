"""N > 1 path of bench.py on CPU: bench.run end to end under world-size-2 gloo with a stub context (rank plan, CPU binding,
barriers, timed steps, MAX over ranks, per-rank report), and the path / map sharding helpers."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubCtx:
    """Stands in for dg.Context: every call bench.run makes, with sleeps instead of GPU work (rank 1 is the slow one)."""

    def __init__(self, rank, n_slots, batch):
        self.rank, self.n_slots, self.batch = rank, n_slots, batch
        self.host_threads = 3
        self.submits = self.replays = self.prepares = 0
        self.frame_bytes = 12

    def submit(self, slot, views):
        import time
        time.sleep(0.002 * (1 + self.rank))
        self.submits += 1

    def wait(self, slot):
        pass

    def prepare(self, slot, views):
        self.prepares += 1

    def replay(self, slot):
        self.replays += 1

    def timing(self, slot):
        return {"raster_ms": 0.5, "setup_ms": 0.1, "host_ms": 0.25 * (1 + self.rank), "n_frames": self.batch, "n_spans": 1000 * self.batch,
                "list_bytes": 4096 * self.batch, "front_end": 2}

    def fallbacks(self):
        return {"front_end": 0, "redone_frames": 0}

    def close(self):
        pass


class _StubBackend:
    def __init__(self, args, device):
        self.args = args

    def load(self, map_id, path_seed, camera="path"):
        import bench
        self.map_seed, self.heavy, self.path_seed, self.camera = map_id[0], map_id[1], path_seed, camera
        B = self.args.batch
        self.n_slots = max(1, min(self.args.slots, (bench.PATH_FRAMES + B - 1) // B))
        self.views = [object()] * self.n_slots
        self.batch_first = [s * B for s in range(self.n_slots)]
        self.ctx = _StubCtx(int(os.environ["RANK"]), self.n_slots, B)
        return self.ctx


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dist.barrier()
    slow = bench.dist_max(0.5 + rank, dist)            # rank 1 is the slow one
    # bench.run end to end (rank plan, CPU binding, barriers, timed steps, MAX over ranks, report gather) on a stub context
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--batch", "250", "--config", "4", "--clock-warmup-ms", "0"])
    holder = {}

    def factory(a, device):
        holder["be"] = _StubBackend(a, device)
        return holder["be"]
    line = bench.run(args, factory)
    be = holder["be"]
    q.put((rank, slow, line, be.map_seed, be.path_seed, be.ctx.submits, len(os.sched_getaffinity(0))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_bench_run():
    import torch.multiprocessing as mp
    import bench
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ncpu = len(os.sched_getaffinity(0))
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 1.5                # MAX over ranks
    line = res[0][2]
    assert res[1][2] is None and line is not None       # only rank 0 reports
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3
    # 3 timed + 1 warmup steps of 4 batches each on every rank
    assert res[0][5] == res[1][5] == 4 * 4
    # value = frames of all ranks / the SLOWER rank's time (rank 1 sleeps 4 ms per batch: 12 batches >= 48 ms)
    assert line["value"] <= 2 * 3 * 1000 / 0.048
    assert line["value"] == pytest.approx(2 * 3 * 1000 / (line["ms_per_step"] * 3 / 1e3))
    assert 0 < line["ms_per_step_min"] <= line["ms_per_step_median"] <= line["ms_per_step_max"]      # the spread over the timed steps
    assert line["ms_per_step_min"] >= 4 * 2.0 * 0.9 and line["latency"] is None                      # rank 0: 4 batches x 2 ms; no latency leg at N > 1
    per = line["per_rank"]
    assert [p["rank"] for p in per] == [0, 1]
    assert [(p["map_seed"], p["path_seed"]) for p in per] == [(1993, 1993), (1994, 1994)] == [(r[3], r[4]) for r in res]
    assert [p["heavy_map"] for p in per] == [False, True] and [p["device"] for p in per] == [0, 1]
    assert line["config"]["baseline_config"] == 4 and "config 4" in line["config"]["workload"]
    assert per[0]["frames_per_s"] > per[1]["frames_per_s"]                  # rank 0 really was faster; the headline is not its rate
    assert per[1]["host_ms_per_batch"] == pytest.approx(0.5)
    if ncpu >= 2:                                                            # each rank pinned itself to its half before starting
        assert res[0][6] + res[1][6] <= ncpu and res[0][6] >= 1


def test_seeded_routes_are_permutations_of_the_same_loop():
    import bench
    route = [(i, i * i) for i in range(37)]
    assert bench.seeded_route(route, 1993) == route
    starts = set()
    for seed in range(1993, 2001):
        rr = bench.seeded_route(route, seed)
        assert sorted(rr) == sorted(route) and len(rr) == len(route)
        starts.add(rr[0])
    assert len(starts) >= 5                             # eight paths, (almost) all entering the loop somewhere else
    assert bench.rank_plan(0, 1) == ((1993, False), 1993)


def test_eight_rank_plans_of_every_config():
    """BASELINE configs -> what each of eight ranks renders: config 4 alternates the light and the heavy map, config 5 is the heavy
    map on every rank, the others the light one; camera paths are always 1993 + rank; sizes and batches divide the path."""
    import bench
    for cfg, (name, W, H, B, maps, camera) in bench.CONFIGS.items():
        plans = [bench.rank_plan(r, 8, cfg) for r in range(8)]
        assert [p[1] for p in plans] == list(range(1993, 2001))
        assert bench.PATH_FRAMES % B == 0 and W % 4 == 0 and f"config {cfg}" in name
        a = bench.parse_args(["--config", str(cfg)])
        assert (a.width, a.height, a.batch) == (W, H, B)
    assert [bench.rank_plan(r, 8, 4)[0] for r in range(4)] == [(1993, False), (1994, True), (1993, False), (1994, True)]
    assert {bench.rank_plan(r, 8, 5)[0] for r in range(8)} == {(1994, True)}
    assert {bench.rank_plan(r, 8, 3)[0] for r in range(8)} == {(1993, False)}
    assert bench.CONFIGS[5][1:3] == (2560, 1600) and bench.CONFIGS[1][5] == "start"
    with pytest.raises(SystemExit):
        bench.parse_args(["--batch", "300"])            # does not divide the 1000-frame path


def test_device_ordinal_survives_masked_visibility():
    """A launcher that gives every rank ONE visible GPU (HIP_VISIBLE_DEVICES per rank): LOCAL_RANK 5 must use ordinal 0."""
    import bench
    assert [bench.pick_device(r, 8) for r in range(8)] == list(range(8))
    assert [bench.pick_device(r, 1) for r in range(8)] == [0] * 8
    assert bench.pick_device(3, 0) == 3                 # no GPU visible (CPU rehearsal): unchanged, dg_create reports it
    assert [bench.pick_device(r, 8, 4) for r in range(4)] == [0, 1, 2, 3]
    with pytest.raises(SystemExit):                     # 8 ranks on 4 unmasked GPUs: two ranks would share a GPU silently
        bench.pick_device(5, 4, 8)
    with pytest.raises(SystemExit):
        bench.pick_device(1, 4, 8)


def _fake_eight_gpu_sysfs(root, numa_of_gpu, cpulists, gpu_first=False):
    """A KFD topology + PCI + NUMA tree shaped like an 8-GPU MI300-class host: two CPU nodes (simd_count 0) and eight GPU nodes whose
    `properties` carry location_id / domain among other two-token lines (and a line that is not two tokens), the PCI device of each
    with its numa_node, the NUMA nodes' cpulists."""
    nodes = os.path.join(root, "class/kfd/kfd/topology/nodes")
    kinds = ["cpu", "cpu"] + ["gpu"] * 8
    if gpu_first:
        kinds = ["gpu"] * 8 + ["cpu", "cpu"]
    g = 0
    for i, kind in enumerate(kinds):
        os.makedirs(os.path.join(nodes, str(i)))
        if kind == "cpu":
            props = "cpu_cores_count 128\nsimd_count 0\nmem_banks_count 1\nlocation_id 0\ndomain 0\n"
        else:
            bus, dom = 0x05 + 0x10 * g, g // 4                       # two PCI domains, four GPUs each
            loc = (bus << 8) | (0 << 3) | 0
            props = f"cpu_cores_count 0\nsimd_count 1024\nmem_banks_count 1\nname\nsimd_arrays_per_engine 1\nlocation_id {loc}\ndomain {dom}\ngfx_target_version 90500\n"
            dev = os.path.join(root, f"bus/pci/devices/{dom:04x}:{bus:02x}:00.0")
            os.makedirs(dev)
            open(os.path.join(dev, "numa_node"), "w").write(f"{numa_of_gpu[g]}\n")
            g += 1
        open(os.path.join(nodes, str(i), "properties"), "w").write(props)
    for n, cl in enumerate(cpulists):
        d = os.path.join(root, f"devices/system/node/node{n}")
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(cl + "\n")


def test_cpu_binding_on_a_made_up_eight_gpu_topology(tmp_path):
    """bench.py pins each rank to CPUs of its GPU's NUMA node before HIP starts.  No multi-GPU box is available to the builder, so the
    sysfs parsing and the split are exercised on a made-up tree with the KFD layout (CPU nodes before or after the GPU nodes, two PCI
    domains, hyper-thread sibling ranges in the cpulists, a GPU without NUMA information, a restricted affinity mask)."""
    import bench
    root = str(tmp_path / "a")
    _fake_eight_gpu_sysfs(root, [0, 0, 0, 0, 1, 1, 1, 1], ["0-63,128-191", "64-127,192-255"])
    assert bench.gpu_numa_nodes(root) == [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = range(256)
    shares = [bench.rank_cpu_share(r, 8, allowed, root) for r in range(8)]
    assert all(len(s) == 32 for s, _ in shares) and all("NUMA node" in how and "4 rank(s)" in how for _, how in shares)
    flat = [c for s, _ in shares for c in s]
    assert sorted(flat) == list(range(256))                          # disjoint, everything used
    assert set(shares[0][0]) <= set(range(0, 64)) | set(range(128, 192)) and set(shares[5][0]) <= set(range(64, 128)) | set(range(192, 256))
    # GPU nodes enumerated before the CPU nodes, GPUs interleaved over the NUMA nodes
    root = str(tmp_path / "b")
    _fake_eight_gpu_sysfs(root, [0, 1, 0, 1, 0, 1, 0, 1], ["0-127", "128-255"], gpu_first=True)
    assert bench.gpu_numa_nodes(root) == [0, 1, 0, 1, 0, 1, 0, 1]
    s1, how = bench.rank_cpu_share(1, 8, allowed, root)
    assert s1 == list(range(128, 160)) and "node 1" in how
    # a launcher that left this rank 16 CPUs in all: fewer CPUs on the node than ranks sharing it -> even split of what is allowed
    s, how = bench.rank_cpu_share(3, 8, range(0, 16), root)
    assert s == [6, 7] and how == "even split of the allowed CPUs"
    # a GPU whose PCI device reports no NUMA node
    root = str(tmp_path / "c")
    _fake_eight_gpu_sysfs(root, [-1] * 8, ["0-255"])
    s, how = bench.rank_cpu_share(2, 8, allowed, root)
    assert s == list(range(64, 96)) and how == "even split of the allowed CPUs"
    # no topology at all (this container): even split; one rank: everything
    assert bench.gpu_numa_nodes(str(tmp_path / "none")) == []
    assert bench.rank_cpu_share(0, 2, range(8), str(tmp_path / "none")) == ([0, 1, 2, 3], "even split of the allowed CPUs")
    assert bench.rank_cpu_share(0, 1, range(8), root)[1] == "all allowed CPUs"


def test_host_threads_respect_the_container_cpu_quota(tmp_path):
    """The GPU boxes show 256 CPUs in the affinity mask and allow 16 CPUs of run time (cgroup v2 cpu.max "1600000 100000"): the default
    thread count must come from the quota, or the process is throttled."""
    import bench
    v2 = tmp_path / "v2"; v2.mkdir()
    (v2 / "cpu.max").write_text("1600000 100000\n")
    assert bench.cgroup_cpu_quota(str(v2)) == 16
    (v2 / "cpu.max").write_text("250000 100000\n")
    assert bench.cgroup_cpu_quota(str(v2)) == 3                       # rounded up
    (v2 / "cpu.max").write_text("max 100000\n")
    assert bench.cgroup_cpu_quota(str(v2)) == 0
    v1 = tmp_path / "v1"; (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("800000\n"); (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cgroup_cpu_quota(str(v1)) == 8
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench.cgroup_cpu_quota(str(v1)) == 0
    assert bench.cgroup_cpu_quota(str(tmp_path / "none")) == 0
    n_aff = len(os.sched_getaffinity(0))
    (v2 / "cpu.max").write_text("200000 100000\n")
    assert bench.default_host_threads(1, str(v2)) == min(n_aff, 2)
    assert bench.default_host_threads(1, str(tmp_path / "none")) == min(n_aff, 16)
    (v2 / "cpu.max").write_text("12800000 100000\n")                  # an 8-GPU node's container: 128 CPUs for 8 ranks
    assert bench.default_host_threads(8, str(v2)) == min(n_aff, 16)
    (v2 / "cpu.max").write_text("1600000 100000\n")                   # two ranks rehearsing on the one-GPU box: 8 each
    assert bench.default_host_threads(2, str(v2)) == min(n_aff, 8)


def test_route_from_wad_gives_a_closed_walk(synth, campath_mod):
    """bench.py --wad: any map yields a route (midpoints of its two-sided lines around their centroid, from the FIRST marker)."""
    for quirks in (False, True):
        wad = synth.build_synth_iwad(1993, quirks=quirks)
        r = campath_mod.route_from_wad(wad, "e1m1")
        assert 8 <= len(r) <= 48 and len(set(r)) > 4
        path = campath_mod.make_camera_path(r, lambda x, y, d: 0.0, 50)
        assert path.shape == (50, 8) and np.isfinite(path).all()


def test_issue_roofline_from_the_committed_counters(tmp_path):
    """`roofline.issue`: wave-instructions per (column, 64-row) chunk by class from profiles/issue.json and the share of the issue capacities
    they fill at a given launch time; tools/issue_counters.py writes that file from rocprofv3 --pmc csv output (checked on a synthetic one)."""
    import csv
    import json
    import subprocess
    import bench
    r = bench.issue_roofline(1280, 800, 1000, 1.8e-3)                 # the committed profile of the default command
    assert r and "error" not in r, r
    pc = r["per_chunk"]
    assert r["chunks_per_launch"] == 1000 * 20 * 64 * 13
    assert 40 < pc["valu"] < 80 and 20 < pc["salu"] < 60 and 0 < pc["valu_paired"] < pc["valu"]
    assert abs(r["per_chunk_total"] - sum(pc[k] for k in ("valu", "salu", "lds", "vmem", "smem"))) < 1e-9
    assert 0.3 < r["vector_issue_slots_filled"] < 1.0 and 0.2 < r["scalar_issue_filled"] < 1.0 and 0.1 < r["lds_cycles_filled"] < 1.0
    assert bench.issue_roofline(123, 45, 6, 1e-3) is None               # a size nobody profiled
    # the extractor on a synthetic counter dump: two launches of 4 frames (the largest grid), one smaller launch ignored
    d = tmp_path / "pmc_x" / "run"
    d.mkdir(parents=True)
    cols = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size",
            "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    vals = {"SQ_INSTS_VALU": 1000.0, "SQ_ACTIVE_INST_VALU2": 250.0, "SQ_INSTS_SALU": 600.0, "SQ_INSTS_LDS": 100.0, "SQ_INSTS_VMEM_RD": 30.0, "SQ_INSTS_VMEM_WR": 4.0,
            "SQ_INSTS_SMEM": 3.0, "SQ_LDS_IDX_ACTIVE": 350.0, "SQ_BUSY_CU_CYCLES": 256.0 * 2000.0}
    with open(d / "1_counter_collection.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(cols)
        for disp, grid in ((1, 4096), (2, 8192), (3, 8192)):
            for name, v in vals.items():
                w.writerow([disp, disp, "Agent 2", 1, 1, 1, grid, 7, "dg::dg_raster_tiles(dg::RasterParams)", 512, 0, 0, 64, 0, 80, name, v * (2 if grid == 4096 else 1), 1000, 2000])
    out = tmp_path / "issue.json"
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "tools", "issue_counters.py"), str(tmp_path), "64x64x4", "--out", str(out)])
    e = json.load(open(out))["64x64x4"]
    assert (e["valu"], e["valu_paired"], e["salu"], e["vmem"], e["launches"], e["grid"]) == (1000.0, 250.0, 600.0, 34.0, 2, 8192)
    assert abs(e["clock_ghz"] - 2.0) < 1e-9                               # 2 000 cycles per CU in 1 000 ns


def test_synth_map_runs_are_labelled_as_not_a_baseline_configuration(monkeypatch):
    """`--synth-map SEED:COLSxROWS:THINGS` (the doom2-scale map of the GPU tier through the whole timed path) is not one of BASELINE.json's
    configurations: the line says so in `config.workload`, carries no side legs, and the flag reaches the backend unchanged."""
    import bench
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-resident", "--no-host-frames", "--no-latency",
                             "--config", "3", "--synth-map", "2002:32x24:500", "--front-end", "segs"])
    holder = {}

    def factory(a, device):
        holder["be"] = _StubBackend(a, device)
        return holder["be"]
    line = bench.run(args, factory)
    assert holder["be"].args.synth_map == "2002:32x24:500"
    assert line["config"]["workload"].startswith("NOT a BASELINE configuration: generated map 2002:32x24:500")
    assert "synthetic IWAD 2002:32x24:500" in line["config"]["workload"] and line["side_legs"] is None
    assert line["n_gpus"] == 1 and line["steps"] == 2
    # the GPU-clock warm-up: untimed steps before the W warm-up steps, for the time asked, reported in the line; the K timed steps stay K
    assert line["gpu_clock_warmup"]["ms"] == 60.0 and line["gpu_clock_warmup"]["steps"] >= 1
    assert holder["be"].ctx.submits == (line["gpu_clock_warmup"]["steps"] + 1 + 2) * 1          # one 1 000-frame batch per step: clock warm-up + W + K
    plain = bench.run(bench.parse_args(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-resident", "--no-host-frames", "--no-latency"]), factory)
    assert plain["config"]["workload"].startswith("BASELINE config 3")
