"""CPU: the N>1 path of bench.py (receiver sharding + barrier + max-over-ranks) with world_size 2 on gloo."""
import socket

import pytest
import torch.multiprocessing as mp

import mp_worker
from ais_catcher_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=mp_worker.worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat, elapsed, value = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert flat == list(range(512))                      # disjoint, complete, ordered by rank
    assert elapsed == pytest.approx(0.020)               # the slowest rank defines the job time
    assert value == pytest.approx(256 * 786432 * 2 * 10 / 0.020 / 1e6)
    assert [shard.owner_of(i, 256) for i in (0, 255, 256, 511)] == [0, 0, 1, 1]
