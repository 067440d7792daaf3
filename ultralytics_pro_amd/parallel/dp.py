"""Batch-sharded data parallelism for the inference / validate path: one process per GPU, weights replicated, images
sharded, NO collective inside the forward path.  The only exchanges are end-of-run ones:

* `max_over_ranks`  - the bench's step time (max over ranks), one scalar all-reduce;
* `gather_detections` - the validator's end-of-run gather of per-image detections to every rank. The reference pickles
  Python lists through `dist.gather_object` (models/yolo/detect/val.py:225-240); here the NMS kernel already produces
  fixed-shape `(B, max_det, 6)` + counts tensors, so it is two `all_gather_into_tensor` calls on RCCL (backend "nccl"
  on ROCm; "gloo" in the CPU tests), no pickling, no host round trip.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the default group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local


def shard_first_image(rank: int, per_rank_batch: int) -> int:
    """Index of the first image of `rank`'s shard in the global (procedural) image stream: weak scaling keeps the
    per-GPU batch fixed (trainer.py:317 divides the global batch instead; `bench.py` reports `scaling: weak`)."""
    return rank * per_rank_batch


def flat_parameter_layout(model):
    """Order of the trainer's flat f32 parameter / gradient buffers, [biases | decayed weights | norm weights] (the reference's three
    optimizer groups, engine/trainer.py:674-682 `build_optimizer`): (groups [(start, numel, has_weight_decay)], meta [(state_dict name,
    flat offset, numel)] in flat order).  Pure bookkeeping on the module tree - no tensors are touched - so every rank of a job (and the
    CPU dry run of `bench.py --gpus N`) derives the same table."""
    import torch.nn as nn
    g0, g1, g2 = [], [], []
    norm = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for mname, mod in model.named_modules():
        for pname, p in mod.named_parameters(recurse=False):
            if not p.requires_grad:
                continue
            full = f"{mname}.{pname}" if mname else pname
            (g2 if "bias" in full else g1 if isinstance(mod, norm) else g0).append(p)
    names = {id(p): n for n, p in model.named_parameters()}
    groups, meta, off = [], [], 0
    for params, wd in ((g2, False), (g0, True), (g1, False)):
        start = off
        for p in params:
            meta.append((names.get(id(p), ""), off, p.numel(), p))
            off += p.numel()
        groups.append((start, off - start, wd))
    return groups, meta


def gradient_bucket_table(meta, groups, target_bytes: int = 16 << 20):
    """Layer spans, last layer first, of at least `target_bytes` of f32 gradients each (DDP's bucket_cap_mb idea; yolov8s: three spans of
    its 44.7 MB), as [(first layer of the span, [flat ranges])].  `meta` = (name, offset, numel, ...) rows in flat order, `groups` =
    (start, numel, ...) rows: within each optimizer group the parameters lie in layer order, so a span of layers is ONE contiguous range
    per group.  A span is complete - every gradient kernel of its layers enqueued - once the backward walk has passed its first layer."""
    per_layer = {}
    for row in meta:
        name, off, n = row[0], row[1], row[2]
        parts = name.split(".")
        li = int(parts[1]) if len(parts) > 1 and parts[0] == "model" and parts[1].isdigit() else -1
        per_layer.setdefault(li, []).append((off, off + n))
    layers = sorted(per_layer, reverse=True)
    spans, cur, cur_bytes = [], [], 0
    for li in layers:
        cur.append(li)
        cur_bytes += 4 * sum(b - a for a, b in per_layer[li])
        if cur_bytes >= target_bytes:
            spans.append(cur)
            cur, cur_bytes = [], 0
    if cur:
        spans.append(cur)
    out = []
    for sp in spans:
        ranges = []
        for g in groups:  # merge the span's parameters of one group into one range
            gstart, gn = g[0], g[1]
            inside = [(a, b) for li in sp for a, b in per_layer[li] if gstart <= a < gstart + gn]
            if inside:
                ranges.append((min(a for a, _ in inside), max(b for _, b in inside)))
        out.append((min(sp), ranges))
    return out


def max_over_ranks(seconds: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_detections(out: torch.Tensor, counts: torch.Tensor):
    """All-gather fixed-shape NMS outputs: (B, max_det, 6), (B,) per rank -> (world*B, max_det, 6), (world*B,)
    ordered by rank, i.e. by global image index for shards made with `shard_first_image`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return out, counts
    world = dist.get_world_size()
    g_out = torch.empty((world * out.shape[0], *out.shape[1:]), dtype=out.dtype, device=out.device)
    g_cnt = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(g_out, out.contiguous())
    dist.all_gather_into_tensor(g_cnt, counts.contiguous())
    return g_out, g_cnt


def gather_ragged(rows: torch.Tensor, counts: torch.Tensor):
    """`gather_detections` for shards that need NOT agree in shape: rank r holds (I_r, W_r, ...) rows + (I_r,) counts with its
    own image count I_r (uneven shards) and its own padded row width W_r (e.g. gt classes padded to that rank's largest
    image).  The ranks first agree on max I and max W (one MAX all-reduce of two integers), pad to that, all-gather, and the
    padding images are dropped again, so the result is ordered by rank like `gather_detections`' and every collective has
    the same shape on every rank (an RCCL all-gather with differing shapes hangs or corrupts)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rows, counts
    world = dist.get_world_size()
    dims = torch.tensor([rows.shape[0], rows.shape[1]], dtype=torch.int64, device=rows.device)
    mx = dims.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    imax, wmax = int(mx[0]), int(mx[1])
    pad = torch.zeros((imax, wmax, *rows.shape[2:]), dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0], :rows.shape[1]] = rows
    cpad = torch.zeros((imax,), dtype=counts.dtype, device=counts.device)
    cpad[:counts.shape[0]] = counts
    nimg = torch.empty((world,), dtype=torch.int64, device=rows.device)
    dist.all_gather_into_tensor(nimg, dims[:1].contiguous())
    g_rows, g_cnt = gather_detections(pad, cpad)
    if bool((nimg == imax).all()):
        return g_rows, g_cnt
    keep = torch.cat([torch.arange(r * imax, r * imax + int(k), device=rows.device) for r, k in enumerate(nimg.tolist())])
    return g_rows[keep], g_cnt[keep]


def _multi_rank() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of `t` from rank `src` (identity on one rank).  Used for the replicate steps of SURVEY 8e: the EMA model's
    float buffers before a sharded validate (engine/trainer.py:695-698)."""
    if _multi_rank():
        dist.broadcast(t, src=src)
    return t


def broadcast_flag(flag: bool, src: int = 0, device=None) -> bool:
    """Rank `src`'s boolean on every rank: the early-stopping flag of the epoch loop (engine/trainer.py:505-508).  The reference pickles
    a one-element list through `dist.broadcast_object_list`; one int32 element says the same without pickling."""
    if not _multi_rank():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.broadcast(t, src=src)
    return bool(int(t.item()))


def reduce_mean_(t: torch.Tensor, dst: int = 0) -> torch.Tensor:
    """`dist.reduce(t, dst, op=AVG)` (engine/validator.py:243-245: the validation loss averaged over the ranks, meaningful on `dst`
    only).  Written as SUM + one division on `dst` - gloo has no AVG, and on RCCL the two are the same arithmetic."""
    if _multi_rank():
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            t /= dist.get_world_size()
    return t


def allreduce_gradients_(flat_grads: torch.Tensor) -> torch.Tensor:
    """The one exchange step of the batch-DP training step: in-place SUM all-reduce of the flat f32 gradient
    buffer (RCCL on the GPU, gloo in the CPU tests).  The reference multiplies each rank's loss by world_size and lets
    DDP AVERAGE the gradients (engine/trainer.py:424-425): every rank ends up with sum_r d(loss_r)/dw - exactly the SUM
    of the unscaled per-rank gradients, which is what this computes (one collective, no scaling pass)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return flat_grads


class BucketedAllReduce:
    """The gradient exchange of the batch-DP training step issued in BUCKETS while backward is still running (the reference's
    DDP overlaps 25 MB buckets with backward, engine/trainer.py:305).  A bucket is a list of [start, end) ranges of the flat
    f32 gradient buffer - the parameters of a span of layers in each of the three optimizer groups - and is reduced (SUM, the
    same exchange as `allreduce_gradients_`) the moment the last gradient kernel of that span has been enqueued:

        issue(k, events)   comm stream waits for `events` (the streams that produced the span's gradients), then every range of
                           bucket k is all-reduced asynchronously (RCCL runs it on its own stream, ordered after the comm stream)
        wait()             the CURRENT stream waits for every issued bucket (stream-level wait, the host does not block);
                           returns the number of buckets that were in flight

    SUM all-reduces of disjoint ranges equal one all-reduce of the whole buffer bit for bit (tests/test_dp_gloo.py).  On CPU
    tensors (gloo, the tests) there are no streams: `issue` starts the asynchronous collectives, `wait` blocks on them."""

    def __init__(self, flat: torch.Tensor, buckets):
        self.flat = flat
        self.buckets = [[(int(a), int(b)) for a, b in rs if b > a] for rs in buckets]
        self.pending = []
        self.issued = set()
        self.comm = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None

    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def issue(self, k: int, events=()):
        if not self.active() or k in self.issued:
            return
        self.issued.add(k)
        if self.comm is not None:
            for ev in events:
                self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                for a, b in self.buckets[k]:
                    self.pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
        else:
            for a, b in self.buckets[k]:
                self.pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def wait(self) -> int:
        n = len(self.issued)
        for w in self.pending:
            w.wait()
        self.pending, self.issued = [], set()
        return n

    def covered(self) -> int:
        return sum(b - a for rs in self.buckets for a, b in rs)
