"""Worker of tests/test_multiproc.py (a separate module so that a spawned interpreter can import it without the
pytest conftest having run)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    import _pkg
    _pkg.load()
    import torch.distributed as dist
    from ais_catcher_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = list(shard.receiver_range(rank, world, 256))
    gathered = [None] * world
    dist.all_gather_object(gathered, ids)  # every rank reports the ids it owns; rank 0 checks the partition
    dist.barrier()
    elapsed = shard.max_over_ranks(0.010 * (rank + 1), dist)  # rank 1 is the slow one
    if rank == 0:
        flat = [i for part in gathered for i in part]
        q.put((flat, elapsed, shard.aggregate_msamples(256 * 786432, world, 10, elapsed)))
    dist.barrier()
    dist.destroy_process_group()
