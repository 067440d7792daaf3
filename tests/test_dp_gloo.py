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


def _ragged_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ultralytics_pro_amd.parallel import gather_ragged, init_distributed

    init_distributed("gloo")
    # rank 0: 3 images, gt width 4; rank 1: 2 images, one of them with 70 boxes (wider padding than rank 0's)
    imgs, width = (3, 4) if rank == 0 else (2, 70)
    gcls = torch.arange(imgs * width, dtype=torch.float32).view(imgs, width) + 1000 * rank
    ngt = torch.tensor([1, 2, 3][:imgs] if rank == 0 else [70, 5], dtype=torch.int32)
    g, n = gather_ragged(gcls, ngt)
    assert g.shape == (5, 70) and n.tolist() == [1, 2, 3, 70, 5]
    assert torch.equal(g[:3, :4], torch.arange(12, dtype=torch.float32).view(3, 4)) and float(g[:3, 4:].abs().sum()) == 0.0
    assert torch.equal(g[3:], torch.arange(140, dtype=torch.float32).view(2, 70) + 1000)
    # 3-D rows (the detection statistics) with uneven image counts only
    rows = torch.full((imgs, 6, 12), float(rank + 1))
    cnt = torch.full((imgs,), rank + 4, dtype=torch.int32)
    gr, gc = gather_ragged(rows, cnt)
    assert gr.shape == (5, 6, 12) and gc.tolist() == [4, 4, 4, 5, 5]
    assert float(gr[:3].min()) == 1.0 and float(gr[3:].min()) == 2.0
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_world2_gloo_ragged_gather_uneven_shards_and_gt_width():
    """engine/validator.gather_stats: ranks with different image counts and different padded gt widths (an image with more
    than max_gt boxes on one rank only) must still issue equal-shape collectives and get the same ordered result."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ultralytics_pro_amd.parallel import BucketedAllReduce, allreduce_gradients_, init_distributed
    from ultralytics_pro_amd.utils import procedural as P

    init_distributed("gloo")
    n = 100_003
    g = P.uniform(f"bucketgrad{rank}", (n,), -3.0, 3.0) * (10.0 ** (P.uniform(f"bucketexp{rank}", (n,), -6, 6)))  # wide range
    whole = allreduce_gradients_(g.clone())
    # three "layer spans", each with one range in each of three "optimizer groups" [0, 1000) | [1000, 90000) | [90000, n)
    buckets = [[(700, 1000), (60000, 90000), (97000, n)], [(300, 700), (20000, 60000), (93000, 97000)],
               [(0, 300), (1000, 20000), (90000, 93000)]]
    flat = g.clone()
    b = BucketedAllReduce(flat, buckets)
    assert b.covered() == n
    b.issue(0)
    b.issue(0)  # issuing a bucket twice must not reduce it twice
    b.issue(1)
    assert b.wait() == 2
    b.issue(2)
    b.wait()
    assert torch.equal(flat, whole)  # bit for bit: SUM over disjoint ranges == SUM over the whole buffer
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_world2_gloo_bucketed_allreduce_equals_single_allreduce():
    """engine/trainer.enable_overlapped_allreduce: the gradient buffer exchanged as per-span buckets (issued while backward
    runs) must equal the single SUM all-reduce bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def _train_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import copy

    from oracle import tasks as ot
    from oracle.loss import v8_detection_loss
    from ultralytics_pro_amd.parallel import allreduce_gradients_, init_distributed, shard_first_image
    from ultralytics_pro_amd.utils import procedural as P

    torch.set_num_threads(2)
    init_distributed("gloo")
    B, sz = 2, 64
    m = ot.DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    m.train()
    first = shard_first_image(rank, B)
    batch = {"img": P.synthetic_images(B, h=sz, w=sz, first=first), **P.synthetic_labels(B, first=first)}
    params = [p for p in m.parameters() if p.requires_grad]
    # (a) the build's exchange: per-rank gradients of loss.sum(), one SUM all-reduce of the flat buffer
    loss, _ = v8_detection_loss(m(batch["img"]), batch, m.stride)
    loss.sum().backward()
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    mine = allreduce_gradients_(flat.clone())
    # (b) the reference's scheme: loss * world_size, DDP averages the gradients (trainer.py:424-425)
    m2 = copy.deepcopy(m)
    for p in m2.parameters():
        p.grad = None
    loss2, _ = v8_detection_loss(m2(batch["img"]), batch, m2.stride)
    (loss2.sum() * world).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m2.parameters() if p.requires_grad])
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
    ref /= world
    assert torch.allclose(mine, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))
    # every rank holds the same reduced gradient, and it differs from the local one (the shards differ)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    assert not torch.allclose(mine, flat)
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_world2_gloo_gradient_exchange_matches_reference_ddp_scheme():
    """SURVEY 8e training row: sum-all-reduce of unscaled per-rank gradients == the reference's (loss * world_size,
    DDP average) on two ranks with different shards (oracle model + loss on CPU, gloo)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_bench_gpus_flag_launches_ranks_itself():
    """`python bench.py --gpus 2` (no launcher around it) starts two ranks through torch.distributed.run as a child
    process (the reference's idiom, utils/dist.py:77-104) and rank 0 prints ONE JSON line with n_gpus = 2.  --dry-run keeps
    the GPU out of it: gloo rendezvous on 127.0.0.1, barrier-bracketed region, MAX over ranks."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for wl, scaling in (("infer", "weak"), ("train", "weak"), ("infer", "strong"), ("val", "weak"), ("train", "strong")):
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run",
                            "--workload", wl, "--scaling", scaling], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        rec = json.loads(lines[0])
        gb = 64 if scaling == "weak" else 32  # strong scaling: the global batch stays 32, 16 per rank
        assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["config"]["global_batch"] == gb
        assert rec["scaling"] == scaling and rec["config"]["per_gpu_batch"] == gb // 2
        assert rec["rccl_ranks_seen"] == 2  # a SUM all-reduce of one 1 per rank went through the process group
        assert rec["ms_per_step"] >= 2.0  # the slower rank (2 ms per step) sets the time
        if wl == "val":  # both ranks' statistics reached the AP integration (rank 0: 2 x 32 x (5 + 6) rows, rank 1: (6 + 7))
            assert rec["images_validated"] == 128 and rec["detection_rows_gathered"] == 32 * (5 + 6 + 6 + 7)


def test_bench_world8_dry_run_of_config3_and_validate():
    """The shape BASELINE config 3 is defined on - yolov8s, global batch 256 over 8 ranks (trainer.py:317: 32 per rank) - and the
    validate path at 8 ranks, executed without a GPU: eight gloo ranks started by `bench.py --gpus 8` itself, shard offsets that tile
    the image stream, the SAME gradient-bucket table on every rank (utils/dist.py:77-104 launches the ranks, trainer.py:424-425 the
    exchange they perform), a bucket-by-bucket SUM all-reduce of a model-sized flat buffer, one JSON line."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    for wl, extra in (("train", ["--model", "yolov8s", "--batch", "256", "--scaling", "strong"]), ("val", ["--batch", "32"])):
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run",
                            "--workload", wl] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 8 and rec["rccl_ranks_seen"] == 8 and rec["config"]["parallelism"] == "dp8"
        assert rec["config"]["per_gpu_batch"] == 32 and rec["config"]["global_batch"] == 256
        assert rec["shard_offsets"] == [32 * k for k in range(8)]
        assert rec["ms_per_step"] >= 8.0  # the slowest rank (8 ms per step) sets the time
        if wl == "train":
            assert rec["scaling"] == "strong" and rec["bucket_tables_identical"] and rec["allreduce_sum_ok"]
            assert rec["gradient_buckets"] == 3 and rec["bucket_coverage"] == rec["gradient_floats"] == 11166544
        else:
            assert rec["images_validated"] == 2 * 32 * 8


def _validator_rank(rank, world, port, golden, q):
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from ultralytics_pro_amd.engine.validator import DetectionValidator
    G = np.load(golden)
    v = DetectionValidator()
    per = 4 // world
    for i in range(rank * per, (rank + 1) * per):  # this rank's shard of the 4 golden images, one image per "batch"
        det = torch.from_numpy(G[f"det{i}"])[None]
        n = det.shape[1]
        out = torch.zeros(1, 300, 6)
        out[:, :n] = det
        gcls = torch.from_numpy(G[f"gt_cls{i}"])
        # GPU-free stand-in for the matching kernel: the reference's own TP matrix of this image
        v._det.append(out)
        v._cnt.append(torch.tensor([n], dtype=torch.int32))
        tp = torch.zeros(1, 300, 10, dtype=torch.uint8)
        tp[0, :n] = torch.from_numpy(G[f"tp{i}"].astype(np.uint8))
        v._tp.append(tp)
        g = torch.zeros(1, 64)
        g[0, : gcls.shape[0]] = gcls
        v._gt.append(g)
        v._ngt.append(torch.tensor([gcls.shape[0]], dtype=torch.int32))
    st = v.get_stats()
    q.put((rank, st["mean"], st["ap"].tolist(), int(st["tp"].shape[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_cpu_baseline_leg_is_wall_clock_bounded():
    """bench.py's `cpu_baseline` leg runs as a child process with a hard limit: on a host it does not own (busy, or torch CPU
    convs oversubscribed at all physical cores) one forward can take minutes, and the default `python bench.py` must still
    finish within minutes.  With a limit shorter than one measurement the parent kills the child and reports that nothing
    was measured instead of waiting."""
    import argparse
    import importlib.util
    import time
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("upa_bench", Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = argparse.Namespace(workload="infer", model="yolov8n", batch=32, imgsz=640, cpu_threads=2)
    t0 = time.perf_counter()
    res = bench.run_cpu_baseline_bounded(args, limit_s=3.0)
    assert time.perf_counter() - t0 < 30.0
    assert res["kind"] == "port" and res["unit"] == "images/s"
    assert res["value"] is None or res.get("truncated")  # nothing (or only a prefix) fits into three seconds


def test_validator_stats_gather_two_ranks_matches_reference_map(golden_dir):
    """End-of-validation path across ranks (models/yolo/detect/val.py:222-240): every rank holds the statistics of its
    image shard, `gather_stats` all-gathers the fixed-shape tensors (gloo here, RCCL on the GPUs) and EVERY rank computes
    the reference's class metrics from the union: mAP / AP table equal tests/golden/map_yolov8n.npz bit for bit."""
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    G = np.load(golden_dir / "map_yolov8n.npz")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_validator_rank, args=(r, 2, port, str(golden_dir / "map_yolov8n.npz"), q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=240) for _ in procs]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    for rank, mean, ap, nrows in res:
        assert nrows == sum(G[f"det{i}"].shape[0] for i in range(4))
        assert np.array_equal(np.asarray(ap), G["ap"]), rank
        assert np.allclose(mean, G["mean"], rtol=0, atol=1e-12), rank


def _replicate_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.engine.validator import DetectionValidator
    from ultralytics_pro_amd.parallel import init_distributed
    init_distributed("gloo")
    # A trainer shell holding only what the replicate steps touch (the flat buffer tensors; building a real trainer needs a GPU):
    # after a multi-rank training step the BatchNorm running statistics differ per rank - rank r's are offset by r here.
    tr = DetectionTrainer.__new__(DetectionTrainer)
    tr.device = torch.device("cpu")
    tr.nbuf = 1000
    base = torch.linspace(-1, 1, tr.nbuf)
    tr.RB = torch.cat([base + 0.25 * rank, torch.zeros(7)])            # (the flat buffer is allowed to be longer than nbuf)
    tr.ERB = torch.cat([0.5 * base + 0.125 * rank, torch.zeros(7)])
    tail_rb, tail_erb = tr.RB[tr.nbuf:].clone(), tr.ERB[tr.nbuf:].clone()
    tr.sync_ema_buffers()
    assert torch.equal(tr.RB[:tr.nbuf], base) and torch.equal(tr.ERB[:tr.nbuf], 0.5 * base)   # every rank now holds rank 0's
    assert torch.equal(tr.RB[tr.nbuf:], tail_rb) and torch.equal(tr.ERB[tr.nbuf:], tail_erb)
    # the early-stopping flag is rank 0's on every rank, whatever the others decided locally (engine/trainer.py:505-508)
    assert tr.broadcast_stop(rank == 0) is True
    assert tr.broadcast_stop(rank != 0) is False
    # the validation loss: accumulated per batch on every rank, averaged over the ranks on rank 0, None elsewhere
    v = DetectionValidator()
    assert v.reduce_loss() is None
    for b in range(3):
        v.add_loss(torch.tensor([1.0, 2.0, 3.0]) * (rank + 1) + b)
    got = v.reduce_loss()
    if rank == 0:
        # rank r accumulates 3 * (r + 1) * [1, 2, 3] + (0 + 1 + 2); mean over ranks, then per batch
        mean_r = sum(3 * (r + 1) for r in range(world)) / world
        want = (mean_r * torch.tensor([1.0, 2.0, 3.0]) + 3.0) / 3
        assert torch.allclose(got, want, rtol=0, atol=1e-6), (got, want)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, float(tr.RB[:tr.nbuf].double().sum()), float(tr.ERB[:tr.nbuf].double().sum())))


def test_world2_gloo_replicate_steps_ema_buffers_stop_flag_validation_loss():
    """SURVEY 8e's "also replicate" clause on two gloo ranks: rank 0's EMA / BatchNorm buffers on every rank before a sharded validate
    (engine/trainer.py:695-698), the stop-flag broadcast (:505-508) and the rank-averaged validation loss (engine/validator.py:243-245)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replicate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1:] == res[1][1:]   # identical buffers on both ranks after the sync
