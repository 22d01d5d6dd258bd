"""N > 1 path of bench.py on CPU: world-size-2 gloo, the sharding helper and the MAX-over-ranks timing reduction."""
import os
import socket
import sys

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
