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


def _bench(args, env=None, launcher=None):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    cmd = [sys.executable] + (launcher or []) + [os.path.join(root, "bench.py")] + args
    p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher environment must start two ranks (one per GPU) and report n_gpus 2 with the
    receivers sharded [0, 255] / [256, 511]; --dry-run exercises exactly that path without touching a GPU."""
    rc, out, err = _bench(["--gpus", "2", "--steps", "5", "--dry-run"])
    assert rc == 0, err
    assert len(out) == 1 and out[0]["n_gpus"] == 2 and out[0]["steps"] == 5 and out[0]["scaling"] == "weak"
    assert out[0]["receiver_ranges"] == [[0, 255], [256, 511]]
    assert out[0]["value"] > 0


def test_bench_under_the_drivers_launcher():
    """The driver's command line: torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 (ranks from the environment)."""
    port = _free_port()
    rc, out, err = _bench(["--gpus", "2", "--steps", "3", "--dry-run"],
                          launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                    "--master-port", str(port)])
    assert rc == 0, err
    assert len(out) == 1 and out[0]["n_gpus"] == 2 and out[0]["receiver_ranges"] == [[0, 255], [256, 511]]


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    rc, out, err = _bench(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not out and "must agree" in err
