"""N > 1 path of bench.py on CPU: world-size-2 gloo, the sharding helper and the MAX-over-ranks timing reduction."""
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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dist.barrier()
    slow = bench.dist_max(0.5 + rank, dist)            # rank 1 is the slow one
    route = bench.rank_route([(i, 2 * i) for i in range(10)], rank, world)
    q.put((rank, slow, route[0], bench.aggregate_fps(1000, world, slow)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 1.5                # MAX over ranks
    assert res[0][3] == res[1][3] == pytest.approx(2000 / 1.5)   # whole-job frames / max time
    assert res[0][2] == (0, 0) and res[1][2] == (4, 8)  # rank 1: rotated by n/2 = 5 -> [5..9,0..4], reversed -> starts at 4


def test_rank_routes_are_permutations_of_the_same_loop():
    import bench
    route = [(i, i * i) for i in range(37)]
    for world in (1, 2, 4, 8):
        for r in range(world):
            rr = bench.rank_route(route, r, world)
            assert sorted(rr) == sorted(route) and len(rr) == len(route)


def test_route_from_wad_gives_a_closed_walk(synth, campath_mod):
    """bench.py --wad: any map yields a route (midpoints of its two-sided lines around their centroid, from the FIRST marker)."""
    for quirks in (False, True):
        wad = synth.build_synth_iwad(1993, quirks=quirks)
        r = campath_mod.route_from_wad(wad, "e1m1")
        assert 8 <= len(r) <= 48 and len(set(r)) > 4
        path = campath_mod.make_camera_path(r, lambda x, y, d: 0.0, 50)
        assert path.shape == (50, 8) and np.isfinite(path).all()
