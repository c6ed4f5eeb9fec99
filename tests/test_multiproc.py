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


def test_eight_ranks_under_the_drivers_launcher():
    """The shape of the driver's 8-GPU run (BASELINE configs[4]: 2048 receivers, 256 per GPU): eight ranks, the ranges
    [0, 255] ... [1792, 2047], one per_rank record each, one line from rank 0 -- so that the first real 8-GPU run cannot fail on plumbing."""
    port = _free_port()
    rc, out, err = _bench(["--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run"],
                          launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                                    "--master-port", str(port)])
    assert rc == 0, err
    assert len(out) == 1 and out[0]["n_gpus"] == 8 and out[0]["steps"] == 20 and out[0]["warmup"] == 5 and out[0]["scaling"] == "weak"
    assert out[0]["receiver_ranges"] == [[256 * r, 256 * r + 255] for r in range(8)]
    assert [p["rank"] for p in out[0]["per_rank"]] == list(range(8)) and all(p["ms_per_step"] > 0 for p in out[0]["per_rank"])
    # weak scaling: the value is the whole job's (eight times one rank's samples over the slowest rank's time)
    assert out[0]["value"] == pytest.approx(8 * 256 * 786432 * 20 / (max(p["ms_per_step"] for p in out[0]["per_rank"]) * 20 * 1e-3) / 1e6, rel=0.05)


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    rc, out, err = _bench(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not out and "must agree" in err
