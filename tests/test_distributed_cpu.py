"""N > 1 path on CPU: two gloo ranks shard a batch by contiguous row blocks and all-gather the packed
results (pypolar_amd.sharding).  The decode itself needs a GPU; here every rank fills its block with a
deterministic function of the global row index so the gathered matrix can be checked exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, rows, nb, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "polar-codes_amd"))
    from pypolar_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_bounds(rows, rank, world)
    idx = torch.arange(lo, hi, dtype=torch.int64)
    out = ((idx[:, None] * 7 + torch.arange(nb)[None, :] * 13) % 251).to(torch.uint8)
    ok = (idx % 3 == 0).to(torch.uint8)
    full = sharding.gather_decoded(sharding.pack_result(out, ok), rows)
    if rows % world == 0:  # the benchmark's fast path: equal blocks, preallocated output, one collective
        buf = torch.empty((rows, nb + 1), dtype=torch.uint8)
        full2 = sharding.gather_decoded(sharding.pack_result(out, ok), equal_rows=True, out=buf)
        assert torch.equal(full2, full)
    gi = torch.arange(rows, dtype=torch.int64)
    exp = torch.cat([((gi[:, None] * 7 + torch.arange(nb)[None, :] * 13) % 251).to(torch.uint8),
                     (gi % 3 == 0).to(torch.uint8)[:, None]], dim=1)
    q.put((rank, bool(torch.equal(full, exp)), lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rows", [64, 65, 3])
def test_two_rank_shard_and_gather(rows):
    world, nb = 2, 22
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rows, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    bounds = sorted((r[2], r[3]) for r in res)
    assert bounds[0][0] == 0 and bounds[-1][1] == rows and bounds[0][1] == bounds[1][0]


def test_shard_bounds_cover_all_rows():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "polar-codes_amd"))
    from pypolar_amd import sharding
    for rows in (0, 1, 7, 8, 1 << 20, (1 << 22) + 5):
        for world in (1, 2, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = sharding.shard_bounds(rows, r, world)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == rows
