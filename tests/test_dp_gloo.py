"""CPU (-m "not gpu"): the N>1 data-parallel plumbing on world_size-2 gloo: disjoint image shards, max-over-ranks
timing, fixed-shape detection gather ordered by global image index."""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ultralytics_pro_amd.parallel import gather_detections, init_distributed, max_over_ranks, shard_first_image
    from ultralytics_pro_amd.utils import procedural as P

    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, world)
    B = 2
    first = shard_first_image(r, B)
    x = P.synthetic_images(B, h=64, w=64, first=first)
    # the shard must equal the matching slice of the global stream (bit-identical on every rank)
    full = P.synthetic_images(B * world, h=64, w=64)
    assert torch.equal(x, full[first: first + B])
    t = max_over_ranks(0.5 + rank)
    assert abs(t - (0.5 + world - 1)) < 1e-12
    out = torch.full((B, 5, 6), float(rank)) + torch.arange(B).view(B, 1, 1)
    counts = torch.tensor([rank + 1, rank + 2], dtype=torch.int32)
    g_out, g_cnt = gather_detections(out, counts)
    assert g_out.shape == (B * world, 5, 6) and g_cnt.tolist() == [1, 2, 2, 3]
    for rr in range(world):
        assert torch.equal(g_out[rr * B: (rr + 1) * B], torch.full((B, 5, 6), float(rr)) + torch.arange(B).view(B, 1, 1))
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_world2_gloo_sharding_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
