#!/usr/bin/env python3
"""bench.py - headline metric of BASELINE.json: images/sec/GPU, YOLOv8n 640x640 bs=32, forward + Detect decode + NMS.

  python bench.py --gpus N --steps K --warmup W [--dtype bf16|f32] [--batch 32]
  (N > 1: launched by torch.distributed.run, one rank per GPU; batch-sharded replicas, no data-path collective)

A "step" = one pass of the hot path over one batch of 32 synthetic images already resident in HBM:
model forward (stem + 63 implicit-GEMM convs + SPPF/upsample kernels) -> Detect decode -> batched NMS
(conf 0.25, iou 0.7, max_det 300: the predict defaults), replayed as ONE hipGraph.
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, timed live with HIP events) and `cpu_baseline`
(the oracle on this host's cores, bounded sample).
"""

from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

GFLOP_PER_IMG = 8.744  # yolov8n @640, 2*MAC over all 64 Conv2d
# the other configs of BASELINE.json (same convention); the headline metric is always yolov8n
GFLOP_OTHER = {"yolov3-tiny": 19.002, "yolov8s": 28.603, "yolov5-BoT3": 7.882 + 0.041, "yolov3-rtdetr": 256.56 + 11.49}
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3  # MFMA f32
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = enough steps for a timed region of >= 1 s (1500 infer / 80 train)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--input-batches", type=int, default=8,
                    help="distinct resident input batches rotated through the steps (8 x 78.6 MB > the 256 MB Infinity Cache: "
                         "the input read of a step is a real HBM read)")
    ap.add_argument("--no-nms-prefilter", action="store_true",
                    help="NMS scans every anchor's scores again instead of the best-class keys the Detect class tails wrote (A/B switch)")
    ap.add_argument("--keep-raw", action="store_true",
                    help="also write Detect's raw per-level maps (the reference's second return value, unused by predict / NMS)")
    ap.add_argument("--full-scores", action="store_true",
                    help="write the (B, nc, A) class scores of the decoded head output as well (default with the NMS prefilter: only the "
                         "boxes and every anchor's best-class key are written - all that single-label NMS reads; identical detections)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default=None, help="default: yolov8n (infer), yolov8s (train)")
    ap.add_argument("--workload", default="infer", choices=["infer", "train", "val"],
                    help="infer = BASELINE config 2 (the headline metric); train = config 3: one batch-DP training step "
                         "(forward, v8DetectionLoss, backward, gradient all-reduce over RCCL, clip + SGD nesterov + EMA); "
                         "val = the validate path: model -> NMS(conf 0.001, multi_label) -> upa_match_predictions per batch shard, "
                         "then the all-gather of the per-image statistics and ap_per_class on every rank")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every GPU keeps --batch images per step (global batch N x --batch); strong = the global "
                         "batch stays --batch and each GPU takes --batch / N of it (the reference's trainer.py:317 "
                         "`batch // world_size`); the JSON line reports which")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--in-flight", type=int, default=0,
                    help="steps in flight: consecutive steps are replayed round-robin on this many HIP streams (each with its "
                         "own graph and static buffers), so the NMS tail of step k overlaps the convolutions of step k+1. "
                         "0 = autotune: a few (in-flight, micro-batches) pairs are timed for 20 steps, the fastest is used")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: run only the multi-rank plumbing (gloo rendezvous, barriers, max-over-ranks, the JSON line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opts", default="",
                    help="dispatch overrides of the model's kernel calls (upa_opts fields, include/upa.h) as name=value,... - A/B "
                         "measurements only, e.g. --opts conv_big_bm=128,c2f32_th=10; `env` = take them from UPA_* variables")
    ap.add_argument("--no-host-results", action="store_true",
                    help="leave the detections in HBM (round-2 behaviour); default: every captured step ends with a kernel that "
                         "copies (B, max_det, 6) rows + counts into pinned host memory, so `value` counts host-visible detections")
    ap.add_argument("--no-mode-dispatch", action="store_true",
                    help="A/B: compile the in-flight copies with the library's default kernels (whole-block c2f64, persistent 3x3) "
                         "instead of the throughput choice of engine/pipeline.py")
    ap.add_argument("--no-kernel-profile", action="store_true",
                    help="skip the per-layer roofline timing (counter-collection runs: tools/pmc_step.sh)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the plan of BASELINE.md section 3 (8, 32, physical cores)")
    ap.add_argument("--cpu-baseline-child", default=None, metavar="OUT.json",
                    help="internal: run only the CPU baseline leg (no GPU is touched) and keep OUT.json up to date after every "
                         "measured line - bench.py starts itself this way as a child process with a wall-clock limit")
    ap.add_argument("--parity-out", default=None, metavar="OUT.pt",
                    help="internal (CPU baseline child): also save the oracle's head output and detections of the headline config's "
                         "batch - the same procedural images as the GPU's first resident batch - for the parent's `parity` object")
    ap.add_argument("--no-parity", action="store_true", help="skip the GPU-vs-oracle comparison of the bench line")
    ap.add_argument("--serial", action="store_true",
                    help="no intra-step concurrency (Detect branches on the main stream): per-kernel durations in a "
                         "rocprofv3 trace of this mode are directly comparable with roofline.avg_launch_us")
    ap.add_argument("--linear-graphs", type=int, default=1,
                    help="with an explicit --in-flight / --micro-batches: 1 = every compiled step is one chain of launches (the "
                         "steps in flight are the only concurrency), 0 = Detect branches fork inside the graph")
    ap.add_argument("--micro-batches", type=int, default=0,
                    help="walk the per-GPU batch as this many concurrent sub-batches (parallel hipGraph branches of ONE "
                         "graph; the whole batch is still processed every step): the latency-bound 20x20/40x40 layers of "
                         "one half overlap the other half's. 0 = autotune together with --in-flight")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, the reference's idiom
    (a generated command + `torch.distributed.run`, ultralytics/utils/dist.py:77-104).  Runs as a CHILD process before
    anything in this process touches the GPU (never exec: replacing a process that initialised HIP takes the box down);
    the child's rank 0 prints the JSON line, this process forwards the return code."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free rendezvous port (dist.py:19-28 does the same)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main_dry_run(args):
    """--dry-run: the multi-rank plumbing of this script without a GPU (gloo): rendezvous, the barrier-bracketed timed
    region, MAX-over-ranks of the time, the rank-0 JSON line.  The step is a host-side stand-in; nothing is measured."""
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    pb = per_rank_batch(args, world)
    extra = {}
    if args.workload == "val":
        # the end-of-run exchange of the validate path on gloo: every rank holds the statistics of ITS shard (stand-in rows: the
        # GPU kernels are not involved), `gather_stats` all-gathers them and every rank integrates the same AP table
        from ultralytics_pro_amd.engine.validator import DetectionValidator
        v = DetectionValidator()
        for bi in range(2):
            out = torch.zeros(pb, 300, 6)
            n = 5 + rank + bi
            out[:, :n, 4] = torch.linspace(0.9, 0.3, n)
            out[:, :n, 5] = torch.arange(n) % 3
            tp = torch.zeros(pb, 300, 10, dtype=torch.uint8)
            tp[:, :n:2] = 1
            v._det.append(out); v._cnt.append(torch.full((pb,), n, dtype=torch.int32)); v._tp.append(tp)
            g = torch.zeros(pb, 4 + 60 * rank)  # ranks deliberately disagree on the padded gt width
            g[:, :3] = torch.tensor([0.0, 1.0, 2.0])
            v._gt.append(g); v._ngt.append(torch.full((pb,), 3, dtype=torch.int32))
        st = v.get_stats()
        extra = {"map50_95": round(float(st["mean"][3]), 6), "images_validated": int(2 * pb * world),
                 "detection_rows_gathered": int(st["tp"].shape[0])}
    # every rank's shard of the image stream: offsets must tile [0, world * pb) in rank order
    from ultralytics_pro_amd.parallel import shard_first_image
    first = shard_first_image(rank, pb)
    offs = [first]
    if world > 1:
        t = torch.zeros(world, dtype=torch.int64)
        t[rank] = first
        dist.all_reduce(t)
        offs = t.tolist()
    extra["shard_offsets"] = offs
    if args.workload == "train":
        # the training step's exchange without a GPU: every rank derives the gradient-bucket table from its own copy of the model
        # (pure bookkeeping, parallel/dp.py) - the tables must be identical or the bucketed all-reduce would pair different ranges -
        # and a SUM all-reduce of a flat f32 buffer of the model's size goes through the process group bucket by bucket
        import hashlib
        from ultralytics_pro_amd.nn.tasks import DetectionModel
        from ultralytics_pro_amd.parallel import flat_parameter_layout, gradient_bucket_table
        groups, meta = flat_parameter_layout(DetectionModel((args.model or "yolov8s") + ".yaml"))
        table = gradient_bucket_table(meta, groups)
        total = groups[-1][0] + groups[-1][1]
        digest = int(hashlib.sha256(repr(table).encode()).hexdigest()[:15], 16)
        same = True
        flat = torch.full((total,), float(rank + 1))
        if world > 1:
            d = torch.zeros(world, dtype=torch.int64)
            d[rank] = digest
            dist.all_reduce(d)
            same = len(set(d.tolist())) == 1
            for _first, ranges in table:
                for a, b in ranges:
                    dist.all_reduce(flat[a:b])
        covered = sum(b - a for _f, ranges in table for a, b in ranges)
        extra.update({"gradient_buckets": len(table), "bucket_tables_identical": same, "gradient_floats": total,
                      "bucket_coverage": covered, "allreduce_sum_ok": bool(torch.all(flat == world * (world + 1) / 2))})
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))  # ranks deliberately uneven: the reported time must be the slowest rank's
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    seen = ranks_seen(dist if world > 1 else None, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work)", "value": round(pb * world * args.steps / dt, 1),
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
                          "vs_baseline": None, "dtype": args.dtype, "data": "none (dry run)", "rccl_ranks_seen": seen,
                          "config": {"workload": f"dry-run of --workload {args.workload}", "global_batch": pb * world,
                                     "per_gpu_batch": pb, "parallelism": f"dp{world}"}, **extra}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def per_rank_batch(args, world: int) -> int:
    """Images per GPU and step: --batch (weak scaling) or --batch / N (strong: trainer.py:317 `batch // world_size`)."""
    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit(f"--scaling strong: the global batch {args.batch} is not divisible by {world} ranks")
        return args.batch // world
    return args.batch


def ranks_seen(dist, dev) -> int:
    """How many ranks the process group's collective actually reached: a SUM all-reduce of one 1 per rank (RCCL on the GPUs,
    gloo in the dry run) - the driver checks it against --gpus."""
    if dist is None:
        return 1
    t = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


_T0 = time.perf_counter()


def _crumb(msg: str) -> None:
    """Progress marker on stderr (stdout carries the one JSON line): tells where a run that never finished was stuck."""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.dry_run:
        return main_dry_run(args)
    if args.model is None:
        args.model = "yolov8s" if args.workload == "train" else "yolov8n"
    if args.steps <= 0:
        args.steps = 80 if args.workload == "train" else (1500 if args.model == "yolov8n" else 200)
    if args.cpu_baseline_child:
        return cpu_baseline_child(args)
    if args.workload == "train":
        return main_train(args)
    if args.workload == "val":
        return main_val(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.nn.modules import conv as pconv
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import procedural as P
    from ultralytics_pro_amd.utils.nms import nms_raw

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = DetectionModel(args.model + ".yaml")
    P.apply_procedural_weights(model)
    model = model.to(dev).eval()
    model.set_compute_dtype(dtype)
    if args.opts:  # every kernel call of this process runs under them (also the per-layer profile's direct replays)
        R.set_default_opts(L.Opts.from_env() if args.opts == "env" else
                           L.Opts(**{k: int(v) for k, v in (kv.split("=") for kv in args.opts.split(","))}))
    if args.serial:
        model.model[-1].concurrent = False
    if not args.keep_raw and hasattr(model.model[-1], "keep_raw"):
        model.model[-1].keep_raw = False  # predict / NMS read only the decoded output; the raw maps stay in registers
        if hasattr(model.model[-1], "nms_keys") and not args.no_nms_prefilter:
            model.model[-1].nms_keys = True   # the class tails also write every anchor's best-class NMS key for the NMS below
            if not args.full_scores and hasattr(model.model[-1], "scores_out"):
                model.model[-1].scores_out = False  # ... and nothing else of the class branch: the NMS below reads boxes + keys only
    # per-rank shard of the global stream: rank r owns batches r*K .. r*K+K-1 (K = --input-batches) of the procedural images
    nin = max(1, args.input_batches)
    pb = per_rank_batch(args, world)
    xs = []
    for j in range(nin):
        xj = P.synthetic_images(pb, first=(rank * nin + j) * pb).to(dev)
        if dtype == torch.bfloat16:
            xj = xj.to(torch.bfloat16)  # the reference's `im.half()` for a half model (predictor.py:151-173)
        xs.append(xj.contiguous())
    x = xs[0]
    if os.environ.get("UPA_POOL_TRACE"):
        print(f"[pool] {x.data_ptr():#x} +{x.numel() * x.element_size():#x} end {x.data_ptr() + x.numel() * x.element_size():#x} input x",
              file=sys.stderr, flush=True)

    rtdetr = "rtdetr" in args.model
    if rtdetr:  # config 5: no NMS, RTDETRPredictor.postprocess (models/rtdetr/predict.py:35-74)
        from ultralytics_pro_amd.utils.nms import rtdetr_postprocess_raw

        def post(o):
            out, counts = rtdetr_postprocess_raw(o[0], 0.25, 300, (args.imgsz, args.imgsz), key="bench")
            return out, counts, None
    else:
        def post(o):
            return nms_raw(o[0], 0.25, 0.7, max_det=300, key="bench")

    if not args.no_host_results:
        # hand the detections to the host inside the captured step: one kernel writes the fixed-shape rows and the counts
        # into pinned host memory (one pair of buffers per compiled copy: keyed by the static device buffer it mirrors)
        post_dev, host_bufs = post, {}

        def post(o):
            out, counts, keep = post_dev(o)
            hb = host_bufs.get(out.data_ptr())
            if hb is None:
                hb = (torch.zeros(out.shape, dtype=out.dtype, pin_memory=True), torch.zeros(counts.shape, dtype=counts.dtype, pin_memory=True))
                host_bufs[out.data_ptr()] = hb
            st = L.current_stream(out.device)
            L.check(L.lib().upa_results_to_host(out.data_ptr(), counts.data_ptr(), out.shape[0], out.shape[1], out.shape[2] * out.element_size(),
                                                hb[0].data_ptr(), hb[1].data_ptr(), st), "results_to_host")
            return out, counts, keep

    from ultralytics_pro_amd.engine.pipeline import PipelinedRunner, autotune
    tuned = None
    with torch.no_grad():
        if args.serial:
            args.micro_batches, args.in_flight = args.micro_batches or 1, args.in_flight or 1
        if args.micro_batches == 0 or args.in_flight == 0:
            # most promising first: how streams land on the runtime's hardware queues depends on creation order
            cands = [(f, m) for f in ((args.in_flight,) if args.in_flight else (3, 1, 2))
                     for m in ((args.micro_batches,) if args.micro_batches else (2, 1))]
            if not args.in_flight and not args.micro_batches:
                # linear graphs only: the forked forms (2 copies x 2 concurrent sub-batches 1.5 ms, 1 x 2 1.08 ms against 0.71 /
                # 0.77 for 4 / 3 linear copies in this tuning pass) never win, and captures with forks inside are the one
                # construct that has misbehaved on this runtime (DESIGN "Two rules for graphs in flight"); `--in-flight` /
                # `--micro-batches` / `--linear-graphs 0` still select them explicitly
                cands = [(4, 1, 0, 1), (3, 1, 0, 1)]
            runner, table = autotune(model, xs, post, candidates=cands, mode_dispatch=not args.no_mode_dispatch)
            tuned = {f"in_flight={k[0]},micro_batches={k[1]},lane_priority={k[2]},linear_graphs={k[3]}": round(t * 1e3, 4)
                     for k, t in table.items()}
            args.in_flight, args.micro_batches = runner.in_flight, runner.micro_batches
        else:
            runner = PipelinedRunner(model, xs, post, micro_batches=args.micro_batches, in_flight=args.in_flight,
                                     mode_dispatch=not args.no_mode_dispatch,
                                     linear=bool(args.serial or (args.linear_graphs and args.micro_batches == 1)))
        run = runner.runs[0]
        _crumb("graphs compiled" + (" (autotuned)" if tuned else ""))
        for _ in range(args.warmup):
            runner.step()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            runner.step()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = pb * world * args.steps / dt
    results = run.result if args.micro_batches > 1 else [run.result]
    ndet = [c for (_, counts, _) in results for c in counts.tolist()]
    if not args.no_host_results:  # the host copies the step itself wrote must equal the device results
        for (o_, c_, _) in results:
            hb = host_bufs[o_.data_ptr()]
            cc, oc = c_.cpu(), o_.cpu()
            assert torch.equal(hb[1], cc) and all(torch.equal(hb[0][b, :int(cc[b])], oc[b, :int(cc[b])]) for b in range(cc.numel())), \
                "host-visible detections differ from the device rows"
    seen = ranks_seen(dist, dev)

    roofline, kernels, cpu_baseline, parity, gpu_speed = None, None, None, None, None
    serial_ms = latency_ms = forked_ms = None
    if rank == 0 and not args.serial:
        # the same step with no concurrency at all (one linear graph, one stream): back-to-back ms/step, and the latency of
        # ONE batch from enqueue to host-visible results
        with torch.no_grad():
            one = PipelinedRunner(model, xs[:2], post, micro_batches=1, in_flight=1, linear=True)
            serial_ms = one.measure(steps=60, warmup=6) * 1e3
            lat = []
            for _ in range(20):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                one.step()
                torch.cuda.synchronize(dev)
                lat.append(time.perf_counter() - t1)
            latency_ms = sorted(lat)[len(lat) // 2] * 1e3
            del one
            # the reference's per-stage report (engine/validator.py:253-256 `Speed: ... per image`): the same serial step cut in two -
            # a graph of the model alone (Detect decode included: it is fused into the head's last convs) timed with events on its
            # stream, the rest of the serial step (NMS + the copy of the rows to the host) as the difference
            try:
                det_, conc_ = model.model[-1], getattr(model.model[-1], "concurrent", None)
                if conc_ is not None:
                    det_.concurrent = False  # one chain of launches, as the serial step it is a part of
                try:
                    fwd = model.compile(xs[0], post=None)
                finally:
                    if conc_ is not None:
                        det_.concurrent = conc_
                for _ in range(5):
                    fwd()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(dev)
                e0.record()
                for _ in range(40):
                    fwd()
                e1.record()
                torch.cuda.synchronize(dev)
                fwd_ms = e0.elapsed_time(e1) / 40
                del fwd
                post_ms = max(serial_ms - fwd_ms, 0.0)
                gpu_speed = {"forward_ms_per_batch": round(fwd_ms, 4), "postprocess_ms_per_batch": round(post_ms, 4),
                             "speed": "Speed: %.4fms preprocess, %.4fms inference, %.4fms loss, %.4fms postprocess per image" % (
                                 0.0, fwd_ms / pb, 0.0, post_ms / pb),
                             "how": "one step at a time: hipGraph of the model (forward + fused Detect decode) between HIP events; "
                                    "postprocess = serial_ms_per_step - forward (NMS + detections to pinned host memory)"}
            except Exception as e:  # noqa: BLE001
                print(f"[bench] Speed split skipped: {e}", file=sys.stderr)
            # ... and one step at a time WITH the concurrency a single step has: the six Detect branches as parallel branches of
            # its graph (independent chains on the 80 / 40 / 20-pixel maps: the small ones fill the rounds the large ones leave
            # open).  Forked graphs do not combine with several steps in flight on this runtime (0.89 ms against 0.62), so this is a
            # one-step figure only.
            forked_ms = None
            if not rtdetr:
                try:
                    fk = PipelinedRunner(model, xs[:2], post, micro_batches=1, in_flight=1, linear=False)
                    forked_ms = fk.measure(steps=60, warmup=6) * 1e3
                    del fk
                except Exception as e:  # noqa: BLE001 - an extra measurement must not cost the run its headline line
                    print(f"[bench] forked one-step leg skipped: {e}", file=sys.stderr)
    elif rank == 0:
        serial_ms = ms_per_step
    if rank == 0:
        _crumb("timed region + serial / latency legs done")
        if not args.no_kernel_profile:
            # per-kernel timing of the launches the TIMED region replayed: under the runner's throughput dispatch, if it has one
            mode = getattr(runner, "throughput_opts", None) or {}
            with (R.use_opts(**mode) if mode else contextlib.nullcontext()):
                roofline, kernels = kernel_profile(model, x, dtype, dev, args, pconv, L, post)
            if roofline is not None and mode:
                roofline["dispatch"] = f"as the timed region: upa_opts {mode}"
            _crumb("kernel profile done")
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            import tempfile
            pdir = tempfile.mkdtemp(prefix="upa_parity_")
            ppath = os.path.join(pdir, "oracle.pt") if not args.no_parity else None
            cpu_baseline = run_cpu_baseline_bounded(args, parity_out=ppath)
            _crumb("cpu baseline done")
            if ppath:
                try:
                    parity = gpu_parity(args, dev, ppath, model, xs[0], results, pb)
                except Exception as e:  # noqa: BLE001 - the comparison must never cost the run its throughput line
                    parity = {"error": f"{type(e).__name__}: {e}"}
                _crumb("parity vs the oracle done")
            import shutil
            shutil.rmtree(pdir, ignore_errors=True)
    timed_s = ms_per_step * 1e-3 * args.steps
    if rank == 0 and timed_s < 0.25:
        print(f"[bench] WARNING: the timed region is {timed_s * 1e3:.1f} ms ({args.steps} steps x {ms_per_step:.3f} ms): below 0.25 s a single "
              "runtime hiccup moves `value` by several percent; the default (--steps 0) times >= 1 s", file=sys.stderr)
    if rank == 0:
        line = {
            # `value` is the whole-job total over n_gpus ranks of per-GPU batch 32 (the contract); the per-GPU figure of
            # BASELINE.json's "images/sec/GPU" is `images_per_sec_per_gpu` = value / n_gpus
            "metric": ("images/sec (whole job = n_gpus x per-GPU bs=32) YOLOv8n 640x640 (forward + Detect decode + NMS)"
                       if args.model == "yolov8n" and pb == 32
                       else f"images/sec (whole job) {args.model} {args.imgsz}x{args.imgsz} per-GPU bs={pb} "
                            f"(forward + {'RT-DETR decoder + postprocess' if rtdetr else 'Detect decode + NMS'})"),
            "value": round(value, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "timed_region_s": round(timed_s, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (procedural images + procedural weights, resident in HBM)",
            "rccl_ranks_seen": seen,
            # the "box/cls match vs CPU ref" half of the metric: this run's GPU output against the oracle's on the same images
            "parity": parity,
            "speed": gpu_speed,
            "config": {"workload": f"{args.model} detect 640x640 bs={pb} {args.dtype} inference, 1 hipGraph/step: "
                                   + (("forward+RT-DETR decoder (f32 rows; " + ("bf16-product linears, matrix-core self-attention" if args.dtype == "bf16" else "exact f32") + ")+postprocess(conf .25, max_det 300)") if rtdetr else
                                      "forward+decode+NMS(conf .25, iou .7, max_det 300)"),
                       "micro_batches": args.micro_batches, "intra_step_concurrency": not (args.serial or runner.linear),
                       "steps_in_flight": max(1, args.in_flight), "lane_priority": runner.priority, "linear_graphs": runner.linear, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "autotune_ms_per_step": tuned,
                       "input_batches_rotated": nin, "input_bytes_resident": int(sum(t.numel() * t.element_size() for t in xs)),
                       "detect_raw_maps_written": bool(args.keep_raw),
                       "detect_class_scores_written": bool(getattr(model.model[-1], "scores_out", True) or not getattr(model.model[-1], "nms_keys", False)),
                       "detections_host_visible": not args.no_host_results, "dispatch_opts": args.opts or None,
                       "dispatch_by_mode": ({"steps in flight > 1 (engine/pipeline.py)": runner.throughput_opts,
                                             "serial / one-step-in-flight legs": "library defaults (whole-block c2f64, conv_ws3)"}
                                            if getattr(runner, "throughput_opts", None) else None),
                       "global_batch": pb * world, "per_gpu_batch": pb, "parallelism": f"dp{world} replicas"},
            "images_per_sec_per_gpu": round(value / world, 1),
            "serial_ms_per_step": None if serial_ms is None else round(serial_ms, 4),
            # strict one-batch-at-a-time rate: per-GPU batch / the serial step (no steps in flight, no intra-step concurrency)
            "value_one_step_in_flight": None if serial_ms is None else round(pb / (serial_ms * 1e-3), 1),
            # one step at a time with its Detect branches as parallel graph branches (intra-step concurrency only)
            "one_step_forked_ms": None if forked_ms is None else round(forked_ms, 4),
            "latency_ms_per_batch": None if latency_ms is None else round(latency_ms, 4),
            "detections_per_image_mean": round(sum(ndet) / max(1, len(ndet)), 1),
            "model_tflops": round(value / world * GFLOP_OTHER.get(args.model, GFLOP_PER_IMG) / 1e3, 2),
            "step_roofline": step_roofline(value / world, ms_per_step, args, kernels),
            "roofline": roofline,
            "critical_path": (kernels or {}).get("critical_path") if isinstance(kernels, dict) else None,
            "kernels": kernels,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_val(args):
    """The validate path across N GPUs (reference engine/validator.py:195-260, models/yolo/detect/val.py:168-240):
    every rank owns a contiguous shard of a synthetic validation set (K batches of per-GPU batch images + synthetic labels); a
    step = one batch through model forward -> NMS(conf 0.001, iou 0.7, multi_label, max_det 300: the validator defaults) ->
    `upa_match_predictions` (TP matrices at the 10 IoU thresholds), captured as ONE hipGraph per batch; after the timed steps
    the per-image statistics are all-gathered over RCCL (two fixed-shape collectives) and every rank integrates the AP table."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine.validator import DetectionValidator
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import metrics as M
    from ultralytics_pro_amd.utils import procedural as P
    from ultralytics_pro_amd.utils.nms import nms_raw

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    args.model = args.model or "yolov8n"
    model = DetectionModel(args.model + ".yaml")
    P.apply_procedural_weights(model)
    model = model.to(dev).eval()
    model.set_compute_dtype(dtype)
    model.model[-1].concurrent = False  # linear graphs, one stream per step in flight (as the inference bench)
    if hasattr(model.model[-1], "keep_raw"):
        model.model[-1].keep_raw = False
    pb = per_rank_batch(args, world)
    nb = max(1, args.input_batches)
    v = DetectionValidator(model)
    runs, labels = [], []
    # four compiled steps in flight: the kernel selection of engine/pipeline.py's throughput mode (`--no-mode-dispatch`: library defaults)
    from ultralytics_pro_amd import _lib as L
    if args.opts:
        R.set_default_opts(L.Opts.from_env() if args.opts == "env" else
                           L.Opts(**{k: int(v) for k, v in (kv.split("=") for kv in args.opts.split(","))}))
    given = {} if (not args.opts or args.opts == "env") else dict(kv.split("=") for kv in args.opts.split(","))
    mode = {} if args.no_mode_dispatch else {k: v for k, v in {"c2f": 4, "conv_ws3": 1, "c2f_stream_rows": -1}.items() if k not in given}
    with torch.no_grad(), (R.use_opts(**mode) if mode else contextlib.nullcontext()):
        for j in range(nb):
            first = (rank * nb + j) * pb
            xj = P.synthetic_images(pb, first=first).to(dev)
            xj = (xj.to(torch.bfloat16) if dtype == torch.bfloat16 else xj).contiguous()
            gt, ngt = v.pack_labels(P.synthetic_labels(pb, first=first), pb, (args.imgsz, args.imgsz), dev)
            labels.append((gt, ngt))

            def post(o, gt=gt, ngt=ngt, j=j):
                out, counts, _ = nms_raw(o[0], v.conf, v.iou, multi_label=True, max_det=v.max_det, key=("val", j))
                tp = R.alloc_plain((pb, v.max_det, len(M.IOUV)), torch.uint8, dev, key=("val_tp", j))
                M.match_predictions_batched(out, counts, gt, ngt, out=tp)
                return out, counts, tp
            runs.append(model.compile(xj, post=post))
    lanes = [torch.cuda.Stream(device=dev) for _ in range(args.in_flight or 4)]

    def step(i):
        with torch.cuda.stream(lanes[i % len(lanes)]):
            runs[i % nb]()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # end of run: one pass over the set = the statistics of the K compiled batches; gather + AP on every rank
    for (out, counts, tp), (gt, ngt) in zip((r.result for r in runs), labels):
        v.add_batch_stats(out, counts, tp, gt, ngt)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    st = v.get_stats()
    t_gather = time.perf_counter() - t1
    seen = ranks_seen(dist, dev)
    if rank == 0:
        value = pb * world * args.steps / dt
        print(json.dumps({
            "metric": f"images/sec (whole job) {args.model} {args.imgsz}x{args.imgsz} validate path, per-GPU bs={pb} "
                      "(forward + NMS conf .001 multi_label + TP matching per step; statistics all-gather + AP at the end)",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (procedural images, labels and weights, resident in HBM)",
            "rccl_ranks_seen": seen,
            "config": {"workload": f"{args.model} validate {args.imgsz}x{args.imgsz} bs={pb} {args.dtype}: 1 hipGraph/step = forward + "
                                   "NMS(conf .001, iou .7, multi_label, max_det 300) + upa_match_predictions",
                       "global_batch": pb * world, "per_gpu_batch": pb, "parallelism": f"dp{world} replicas",
                       "val_set": f"{nb} batches per rank = {nb * pb * world} images", "steps_in_flight": len(lanes), "dispatch_by_mode": mode or None, "dispatch_opts": args.opts or None,
                       "exchange": "end of run: all_gather_into_tensor of (rows, counts) and (gt classes, counts) over RCCL"},
            "images_per_sec_per_gpu": round(value / world, 1),
            "gather_and_map_ms": round(t_gather * 1e3, 3),
            "images_in_map": int(len(v._cnt) * pb * world), "detection_rows_gathered": int(st["tp"].shape[0]),
            "map": {"precision": float(st["mean"][0]), "recall": float(st["mean"][1]), "map50": float(st["mean"][2]),
                    "map50_95": float(st["mean"][3])},
            "roofline": None, "cpu_baseline": None}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_train(args):
    """BASELINE config 3: yolov8s 640x640, per-GPU batch 32 (global 256 on 8 GPUs), one training step per `step`:
    train-mode forward, v8DetectionLoss + TaskAlignedAssigner, backward, ONE all-reduce (SUM) of the flat f32 gradient
    buffer over RCCL/xGMI, clip_grad_norm_(10) + SGD(nesterov) + EMA.  Weak scaling: per-GPU work is fixed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    from ultralytics_pro_amd import _lib as L
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.engine.trainer import DetectionTrainer
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import procedural as P

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = DetectionModel(args.model + ".yaml")
    P.apply_procedural_weights(model)
    tr = DetectionTrainer(model, dtype=dtype, device=dev, world_size=world)
    pb = per_rank_batch(args, world)
    x = P.synthetic_images(pb, h=args.imgsz, w=args.imgsz, first=rank * pb).to(dev)
    lab = P.synthetic_labels(pb, first=rank * pb)
    buckets = None
    if world > 1:  # eager steps exchange the gradients in buckets while backward runs (the graph mode keeps one all-reduce)
        buckets = tr.enable_overlapped_allreduce()
    # eager launches (weight gradients overlap the data-gradient chain on a side stream) vs hipGraph replay of the
    # same step: time a few steps of each and keep the faster mode
    def _time(n=4):
        torch.cuda.synchronize(dev)
        t0_ = time.perf_counter()
        for _ in range(n):
            tr.step(x, lab)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0_) / n

    for _ in range(2):
        tr.step(x, lab)
    t_eager = _time()
    tr.compile(x, lab, warm_steps=0)
    tr.step(x, lab)
    t_graph = _time()
    mode = "hipGraph replay" if t_graph <= t_eager else "eager launches"
    if dist is not None:  # every rank must run the same mode (the collectives differ): rank 0 decides
        flag = torch.tensor([1.0 if t_graph <= t_eager else 0.0], device=dev)
        dist.broadcast(flag, 0)
        mode = "hipGraph replay" if float(flag.item()) > 0.5 else "eager launches"
    if mode == "eager launches":
        tr._graphs = None
    tr.allreduce_exposed_ms()  # drop the records of the tuning steps
    for _ in range(max(args.warmup, 1)):
        items = tr.step(x, lab)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        items = tr.step(x, lab)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = pb * world * args.steps / dt
    exposed = tr.allreduce_exposed_ms()
    seen = ranks_seen(dist, dev)
    roofline = cpu_baseline = None
    if rank == 0:
        roofline = wgrad_profile(tr, L, R, dev, dtype, cfg_key=f"{args.model} bs={pb} {args.dtype} train")
        if not args.no_cpu_baseline and world == 1:
            cpu_baseline = run_cpu_baseline_bounded(args)
        nparam = sum(n for _, n, _ in tr.groups)
        print(json.dumps({
            "metric": f"images/sec {args.model} {args.imgsz}x{args.imgsz} training step, per-GPU batch {pb} "
                      "(forward + v8DetectionLoss + backward + gradient all-reduce + clip/SGD-nesterov/EMA)",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (procedural images, labels and initial weights, resident in HBM)",
            "rccl_ranks_seen": seen,
            # time the optimizer waited for the bucketed all-reduce after backward had finished (eager mode, N > 1)
            "allreduce_exposed_ms": None if exposed is None else round(exposed, 4),
            "config": {"workload": f"{args.model} detect {args.imgsz}x{args.imgsz} batch-DP training step, per-GPU batch "
                                   f"{pb}, {args.dtype} activations / f32 master weights and gradients",
                       "global_batch": pb * world, "per_gpu_batch": pb, "parallelism": f"dp{world}",
                       "exchange": (f"all-reduce(SUM) of {nparam * 4 / 1e6:.1f} MB f32 gradients per step (RCCL): "
                                    + (f"{len(buckets)} buckets issued during backward on a communication stream "
                                       f"(first layers {[b[0] for b in buckets]})" if buckets and mode == "eager launches"
                                       else "one collective between the backward and optimizer graphs")),
                       "optimizer": "SGD(lr 0.01, momentum 0.9, nesterov, wd 5e-4) + clip 10.0 + EMA",
                       "batchnorm_statistics": "from the convolutions' own workgroups where the kernel has a statistics epilogue "
                                               "(upa_conv2d_bn_stats: conv_big / conv1x1_stream / conv_ws3), a reduction pass elsewhere",
                       "grad_scaler": bool(tr.scaler.enabled),
                       "execution": mode, "tuning_ms_per_step": {"eager": round(t_eager * 1e3, 3), "graph": round(t_graph * 1e3, 3)}},
            "images_per_sec_per_gpu": round(value / world, 1),
            "loss_items": [round(float(v), 4) for v in items.tolist()],
            # the whole step against the matrix peak: forward + data gradient + weight gradient = 3 x the forward's 2 MAC count
            "step_roofline": {"model_tflops": round(3 * GFLOP_OTHER.get(args.model, GFLOP_PER_IMG) * value / world / 1e3, 1),
                              "frac_of_mfma_peak": round(3 * GFLOP_OTHER.get(args.model, GFLOP_PER_IMG) * value / world / 1e3 / PEAK_BF16_TFLOPS, 4),
                              "algorithmic_gflop_per_image": round(3 * GFLOP_OTHER.get(args.model, GFLOP_PER_IMG), 2)},
            "roofline": roofline, "cpu_baseline": cpu_baseline}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def wgrad_profile(tr, L, R, dev, dtype, reps=5, cfg_key=None):
    """Dominant kernel of the training step = the weight-gradient MFMA kernel: every layer's launch re-issued `reps` times
    back to back on the current stream between HIP events (its operands are still resident from the last step)."""
    fam = {}
    lib = L.lib()
    st = L.current_stream(dev)
    scratch = {}
    for cv in tr.convs:
        if cv.x is None:
            continue
        vx = R.view_of(cv.x)
        oh, ow = (vx.h + 2 * cv.p - cv.k) // cv.s + 1, (vx.w + 2 * cv.p - cv.k) // cv.s + 1
        dz = scratch.setdefault((vx.n, cv.cout, oh, ow), torch.zeros(vx.n, oh, ow, cv.cout, dtype=dtype, device=dev))
        dw = torch.zeros(cv.cout, cv.cin, cv.k, cv.k, device=dev)
        ws = tr.ctx.wgrad_ws

        def call():
            L.check(lib.upa_conv2d_wgrad(vx.ptr, vx.n, vx.h, vx.w, cv.cin, vx.ld, dz.data_ptr(), cv.cout, cv.cout, dw.data_ptr(),
                                         cv.k, cv.s, cv.p, 1, vx.dtype, ws.data_ptr(), ws.numel(), st), "wgrad")
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        bf = dtype == torch.bfloat16
        # the dispatch of upa_conv2d_wgrad (csrc/train.hip): bf16 MFMA kernels where the channel counts allow
        if bf and cv.k == 3 and cv.cout >= 16 and (cv.cin % 8 == 0 or cv.cin < 8):
            # LDS-DMA ring kernels; 9 - 16 input channels keep the register-staged narrow form
            fam_ = "wgrad_bf16_k3_kernel" if 8 < cv.cin <= 16 else "wgrad_k3_ring_kernel"
            name, peak = "void (anonymous namespace)::%s<%d, %d>((anonymous namespace)::WgradParams)" % (
                fam_, cv.s, 16 if cv.cin <= 16 else 64), PEAK_BF16_TFLOPS
        elif bf and cv.k == 1 and cv.s == 1 and cv.p == 0 and cv.cin >= 32 and cv.cout >= 32 and cv.cin % 8 == 0:
            name, peak = "(anonymous namespace)::wgrad_k1_ring_kernel((anonymous namespace)::WgradParams)", PEAK_BF16_TFLOPS
        else:
            small = cv.cin <= 32 or cv.cout <= 32
            mt = 4 if (cv.k == 1 and cv.cin >= 128 and cv.cout >= 128) else (1 if small else 2)
            name = "void (anonymous namespace)::wgrad_kernel<%s, %d, %d, %d>((anonymous namespace)::WgradParams)" % (
                "unsigned short" if bf else "float", mt, mt, cv.k)
            peak = PEAK_F32_TFLOPS
        d = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, peak=peak))
        d["launches"] += 1
        d["ms"] += ms
        d["flops"] += 2.0 * vx.n * oh * ow * cv.cout * cv.cin * cv.k * cv.k
        # algorithmic bytes: the layer input and the output gradient read once (activation dtype), the f32 weight gradient written once
        d["bytes"] += vx.n * (vx.h * vx.w * cv.cin + oh * ow * cv.cout) * (2 if bf else 4) + 4.0 * cv.cout * cv.cin * cv.k * cv.k
    name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
    avg_s = d["ms"] / d["launches"] * 1e-3
    tf = d["flops"] / d["launches"] / avg_s / 1e12
    # HBM bytes per launch of the dominant family from the committed PMC passes (tools/pmc_wgrad.sh: FETCH_SIZE x 2 + WRITE_SIZE,
    # separate passes); only valid for the configuration they were collected on
    traffic, traffic_src = None, None
    for pf in sorted((ROOT / "profiles").glob("r*_pmc_wgrad_summary.json"), reverse=True):
        try:
            pmc = json.loads(pf.read_text())
            if pmc.get("config") == cfg_key and name in pmc["kernels"]:
                traffic = round(pmc["kernels"][name]["hbm_bytes_per_launch"])
                traffic_src = f"profiles/{pf.name} (rocprofv3 --pmc, separate passes)"
                break
        except (OSError, KeyError, ValueError):
            pass
    return {"kernel": name, "bound": "mfma", "achieved": round(tf, 2), "peak": d["peak"], "unit": "TFLOP/s",
            "frac": round(tf / d["peak"], 4), "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": d["launches"],
            "avg_launch_us": round(avg_s * 1e6, 1),
            "algorithmic_flops_per_launch": d["flops"] / d["launches"],
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
            "note": "weight gradient: bf16 MFMA (v_mfma_f32_16x16x32_bf16, operands DMAed into an LDS ring and read with "
                    "ds_read_b64_tr_b16) where channel counts allow, exact-f32 MFMA otherwise; the time includes the "
                    "partial-sum reduction kernel",
            "wgrad_ms_per_step": round(sum(v["ms"] for v in fam.values()), 3),
            "families": {k: dict(launches=v["launches"], avg_us=round(v["ms"] / v["launches"] * 1e3, 1),
                                 tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)) for k, v in sorted(fam.items())}}


def run_cpu_train_baseline(args):
    """The oracle training step (torch autograd on this host's cores), bounded sample: bs 4, 320x320."""
    from oracle import tasks as ot
    from oracle import train as otr
    from ultralytics_pro_amd.utils import procedural as P

    host_cores = os.cpu_count() or 1
    cores = min(host_cores, args.cpu_threads or 32)  # torch CPU convs regress when oversubscribed (see run_cpu_baseline)
    torch.set_num_threads(cores)
    m = ot.DetectionModel(args.model + ".yaml")
    P.apply_procedural_weights(m)
    st = otr.TrainState(m)
    bs, sz = 4, 320
    batch = {"img": P.synthetic_images(bs, h=sz, w=sz), **P.synthetic_labels(bs)}
    otr.train_step(m, st, batch)
    best = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        otr.train_step(m, st, batch)
        best = min(best, time.perf_counter() - t0)
    px_ratio = (sz * sz) / float(args.imgsz * args.imgsz)
    return {"value": round(bs / best * px_ratio, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle train step (torch CPU fp32 autograd) {args.model} bs={bs} {sz}x{sz}, best of 2 = "
                      f"{bs / best:.2f} images/s at {sz}px, scaled by the pixel ratio to {args.imgsz}px; host has "
                      f"{host_cores} logical cores"}


def step_roofline(img_per_s, ms_per_step, args, kernels):
    """The whole step against the chip: MFMA (algorithmic FLOPs of every conv / peak), HBM (algorithmic bytes of every conv,
    each reading its input and writing its output once, / 8 TB/s) and the ratio of the conv HBM floor to the measured step."""
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    tf = img_per_s * GFLOP_OTHER.get(args.model, GFLOP_PER_IMG) / 1e3
    out = {"model_tflops": round(tf, 2), "frac_of_mfma_peak": round(tf / peak, 4)}
    if kernels:
        gb = kernels["conv_algorithmic_bytes"] / 1e9
        out.update({"algorithmic_GB_per_step": round(gb, 4), "algorithmic_GBs": round(gb / (ms_per_step * 1e-3), 1),
                    "frac_of_hbm_peak": round(gb / (ms_per_step * 1e-3) / PEAK_HBM_GBS, 4),
                    "conv_hbm_floor_over_step": round(kernels["conv_hbm_floor_ms"] / ms_per_step, 4)})
    return out


def kernel_profile(model, x, dtype, dev, args, pconv, L, post, reps=10):
    """Device time of every conv launch of one step, measured live with HIP events on the launch stream.

    One eager forward records the launch list; every launch is then captured `reps` times back to back into its own
    hipGraph and the replay is bracketed by events on the stream it runs on, so the figure is kernel time (no host
    launch gaps) and is comparable with rocprofv3's per-kernel AverageNs.  Launches are grouped by the exact kernel
    instantiation name rocprofv3 reports; the roofline is given for the instantiation with the largest total time."""
    from ultralytics_pro_amd.engine import runtime as R
    from ultralytics_pro_amd.nn.modules import block as pblock
    from ultralytics_pro_amd.nn.modules import head as phead

    code = L.dtype_code(dtype)
    es = 2 if code == L.UPA_BF16 else 4
    tname = "unsigned short" if es == 2 else "float"
    calls = []  # (kernel name, flops, algorithmic bytes, replay callable)
    orig = pconv.hip_conv2d
    orig_tail = phead.Detect._tail_call

    def conv_name(n, h, w, cin, pk, stride, pad, act, residual):
        if pk.stem:
            return (f"void stem_mfma_kernel<{pk.cout // 16}, {pk.k}, {stride}, {'true' if act == 1 else 'false'}>(StemParams)" if es == 2 else
                    f"void stem_conv_kernel<{tname}, 16, {'true' if act == 1 else 'false'}>(StemParams)")
        var = L.lib().upa_conv_variant(n, h, w, cin, pk.cout, pk.k, stride, pad, code, R.opts_ptr())
        if (var >> 26) & 1:  # 8-wave two-group phased kernel for the MFMA-bound 3x3 stride-1 layers (conv_p8.hip)
            return "conv_p8_kernel(BigParams)"
        if (var >> 25) & 1:  # 4-wave 32x32x16-MFMA kernel for the MFMA-bound 3x3 layers (conv_mm.hip): <ACT, RES>
            return "void conv_mm_kernel<%d, %s>(MmParams)" % (act, "true" if residual is not None else "false")
        if (var >> 24) & 1:  # persistent weights-stationary 3x3 (conv_ws3.hip): <NT, MT>
            return "void conv_ws3_kernel<%d, %d>(BigParams)" % ((var >> 4) & 15, var & 15)
        if (var >> 23) & 1:  # large-tile LDS-shared-operand kernel (conv_big.hip): <KS, STRIDE, WM, WN, MT, NT>
            ntb, mt = (var >> 4) & 15, 4 if (var & 15) == 2 else 2
            wm, wn, nt = (8, 1, 4) if (ntb == 4 and mt == 4) else (4, 2, ntb // 2)
            if ntb == 4:
                mt = 2
            if ntb == 5:  # 80 output channels: 8 x 1 waves, 5 tiles each, MT = 2 (256 px) or 1 (128 px)
                wm, wn, nt, mt = 8, 1, 5, (2 if (var & 15) == 2 else 1)
            return "void conv_big_kernel<%d, %d, %d, %d, %d, %d, 0>(BigParams)" % (pk.k, stride, wm, wn, mt, nt)
        if (var >> 22) & 1:  # streaming pointwise kernel (conv1x1.hip): <NTW, MT, WAVES, EPI>
            return "void conv1x1_stream_kernel<%d, %d, %d, 0>(C1Params)" % (var & 15, (var >> 4) & 15, (var >> 8) & 31)
        if (var >> 21) & 1 and (var >> 8) & 1:  # 16 -> 16 channel variant of the pipelined kernel: <act, residual>
            return "void conv3x3_c16_kernel<%d, %s>(PipeParams)" % (act, "true" if residual is not None else "false")
        if (var >> 21) & 1:  # software-pipelined 3x3 (conv_pipe.hip): <NTW, act, residual>
            return "void conv3x3_pipe_kernel<%d, %d, %s>(PipeParams)" % (var & 15, act, "true" if residual is not None else "false")
        return "void %s<%s, %d, %d, %d, %d, %d>(ConvParams)" % (
            "conv_ws_kernel" if (var >> 20) & 1 else "conv_igemm_kernel", tname, (var >> 12) & 15,
            (var >> 8) & 15, (var >> 4) & 15, var & 15, (var >> 16) & 15)

    def rec(xx, pk, stride, pad, act, out=None, residual=None, out_dtype=None, key=None, up=None):
        y = orig(xx, pk, stride, pad, act, out=out, residual=residual, out_dtype=out_dtype, key=key, up=up)
        n, cin, h, w = xx.shape
        oh, ow = y.shape[2], y.shape[3]
        flops = 2.0 * n * oh * ow * pk.cout * cin * pk.k * pk.k
        nbytes = n * h * w * cin * xx.element_size() + n * oh * ow * pk.cout * es * (2 if residual is not None else 1) \
            + pk.cout * cin * pk.k * pk.k * es
        if up is not None:  # virtual Upsample + Concat: the leading channels are read at quarter size
            nbytes -= n * h * w * up.channels * xx.element_size() * 3 // 4
        calls.append((conv_name(n, h, w, cin, pk, stride, pad, act, residual), flops, nbytes,
                      lambda: orig(xx, pk, stride, pad, act, out=y, residual=residual, out_dtype=out_dtype, up=up)))
        return y

    def rec_tail(self, t, conv, raw, kind, i, plan):
        orig_tail(self, t, conv, raw, kind, i, plan)
        n, cin, h, w = t.shape
        cout = conv.out_channels
        plan_keep = dict(plan)
        # the fused 1x1 + decode launch (conv1x1.hip EPI 1 / 2): reads t once, writes 4 or nc f32 rows per anchor
        var = L.lib().upa_conv_variant(n, h, w, cin, 64 if kind == 1 else max(16, (cout + 7) // 8 * 8), 1, 1, 0, code, R.opts_ptr())
        name = "void conv1x1_stream_kernel<%d, %d, %d, %d>(C1Params)" % (var & 15, (var >> 4) & 15, (var >> 8) & 31, kind)
        flops = 2.0 * n * h * w * cout * cin
        nbytes = n * h * w * cin * 2 + n * h * w * (4 if kind == 1 else self.nc) * 4 + cout * cin * 2 + \
            (n * h * w * cout * 2 if raw is not None else 0)
        calls.append((name, flops, nbytes, lambda: orig_tail(self, t, conv, raw, kind, i, plan_keep)))

    orig_pair = L.lib().upa_bottleneck_pair
    orig_c2f = L.lib().upa_c2f_fused
    orig_btail = L.lib().upa_detect_branch_tail
    orig_paircv2 = L.lib().upa_bottleneck_pair_cv2
    orig_c2f64 = L.lib().upa_c2f64_fused
    orig_c2f32up = L.lib().upa_c2f32_up_fused
    c2f32up_calls = []
    pair_calls, c2f_calls, btail_calls, paircv2_calls, c2f64_calls = [], [], [], [], []

    class _LibProxy:
        """Forwards every C entry to the real library, recording the fused-block launches (Bottleneck / C2f / Detect call
        them directly, not through hip_conv2d)."""

        def __getattr__(self, name):
            return getattr(real_lib, name)

        def upa_bottleneck_pair(self, *a):
            rc = orig_pair(*a)
            if rc == 0:
                pair_calls.append(a)
            return rc

        def upa_bottleneck_pair_cv2(self, *a):
            rc = orig_paircv2(*a)
            if rc == 0:
                paircv2_calls.append(a)
            return rc

        def upa_c2f_fused(self, *a):
            rc = orig_c2f(*a)
            if rc == 0:
                c2f_calls.append(a)
            return rc

        def upa_c2f64_fused(self, *a):
            rc = orig_c2f64(*a)
            if rc == 0:
                c2f64_calls.append(a)
            return rc

        def upa_c2f32_up_fused(self, *a):
            rc = orig_c2f32up(*a)
            if rc == 0:
                c2f32up_calls.append(a)
            return rc

        def upa_detect_branch_tail(self, *a):
            rc = orig_btail(*a)
            if rc == 0:
                btail_calls.append(a)
            return rc

    real_lib = L.lib()
    proxy = _LibProxy()
    mods = (pconv, pblock, phead)
    pool = R.BufferPool()
    orig_libfn = L.lib
    try:
        for m in mods:
            m.hip_conv2d = rec
        phead.Detect._tail_call = rec_tail
        L.lib = lambda: proxy
        with torch.no_grad(), R.static_buffers(pool):
            post(model._predict_once(x))
    finally:
        for m in mods:
            m.hip_conv2d = orig
        phead.Detect._tail_call = orig_tail
        L.lib = orig_libfn
    for a in pair_calls:  # (x, n, h, w, c, ldx, w1, b1, w2, b2, y, ldy, residual, act, dtype, opts, stream)
        n_, h_, w_, c_ = a[1], a[2], a[3], a[4]
        flops = 2 * 2.0 * n_ * h_ * w_ * c_ * c_ * 9
        nbytes = 2 * (n_ * h_ * w_ * c_ * 2 * (2 + (0.5 if a[12] else 0)) + c_ * c_ * 9 * 2)  # two convs, each in + out (+ residual)
        calls.append(("void conv_pair_kernel<%d, %s, false>(PairParams)" % (c_ // 32, "true" if a[12] else "false"), flops, nbytes,
                      (lambda a=a: orig_pair(*a[:16], L.current_stream(dev)))))
    for a in paircv2_calls:  # (x, y0, n, h, w, ldx, w1, b1, w2, b2, residual, wc_std, wc_b, bc, out, ldout, act, dtype, opts, stream)
        npx = a[2] * a[3] * a[4]
        flops = 2.0 * npx * (2 * 9 * 32 * 32 + 96 * 64)
        nbytes = npx * (64 + 64) * 2 + (18 * 32 * 32 + 96 * 64) * 2  # y0 | y1 in, 64 channels out, weights
        calls.append(("void conv_pair_kernel<1, %s, true>(PairParams)" % ("true" if a[10] else "false"), flops, nbytes,
                      (lambda a=a: orig_paircv2(*a[:19], L.current_stream(dev)))))
    for a in c2f_calls:  # (x, n, h, w, c1, ldx, c, nb, shortcut, w1, b1, wm, bm, w2, b2, y, c2, ldy, act, dtype, opts, stream)
        npx, c1_, c_, nb_, c2_ = a[1] * a[2] * a[3], a[4], a[6], a[7], a[16]
        wts = c1_ * 2 * c_ + nb_ * 18 * c_ * c_ + (2 + nb_) * c_ * c2_
        flops = 2.0 * npx * wts
        nbytes = npx * (c1_ + c2_) * 2 + wts * 2  # block input + block output + weights
        o_ = R.current_opts()
        th = 10 if (c_ != 16 and nb_ == 2 and o_ is not None and o_.c2f32_th == 10) else 16
        stream_form = c_ == 32 and th == 16 and (o_ is None or o_.c2f_stream != 1)  # the line-buffer kernels (csrc/c2f_stream.hip)
        name = ("void c2f16_fused_kernel<%d>(C2fParams)" % (8 if (o_ is not None and o_.c2f16_waves == 8) else 4) if c_ == 16 else
                ("c2f32_stream2_kernel(C2fsParams)" if nb_ == 2 and (o_ is None or o_.c2f_stream != 2) else
                 "void c2f32_stream_kernel<2>(C2fsParams)" if nb_ == 2 else "void c2f32_stream1_kernel<1>(C2fsParams)") if stream_form else
                "void c2f32_fused_kernel<%d, %d>(C2f32Params)" % (nb_, th))
        calls.append((name, flops, nbytes, (lambda a=a: orig_c2f(*a[:21], L.current_stream(dev)))))
    for a in c2f32up_calls:  # (x, n, h, w, c1, ldx, up, up_c, up_ld, nb, shortcut, w1, b1, wm, bm, w2, b2, y, c2, ldy, act, dtype, opts, stream)
        npx, c1_, upc_, nb_, c2_ = a[1] * a[2] * a[3], a[4], a[7], a[9], a[18]
        wts = c1_ * 64 + nb_ * 18 * 32 * 32 + (2 + nb_) * 32 * c2_
        o_ = R.current_opts()
        name = ("void c2f32_stream1_kernel<%d>(C2fsParams)" % (c1_ // 64) if (c1_ <= 192 and (o_ is None or o_.c2f_stream != 1)) else
                "void c2f32_fused_kernel<1, 16, true>(C2f32Params)")
        calls.append((name, 2.0 * npx * wts, npx * (c1_ - upc_ * 3 // 4 + c2_) * 2 + wts * 2,  # block input (the upsampled channels at quarter size) + output + weights
                      (lambda a=a: orig_c2f32up(*a[:23], L.current_stream(dev)))))
    for a in c2f64_calls:  # (x, n, h, w, c1, ldx, up, up_c, up_ld, nb, shortcut, w1, b1, wm, bm, w2, b2, y, c2, ldy, act, dtype, opts, stream)
        npx, c1_, upc_, nb_, c2_ = a[1] * a[2] * a[3], a[4], a[7], a[9], a[18]
        wts = c1_ * 128 + nb_ * 18 * 64 * 64 + (2 + nb_) * 64 * c2_
        flops = 2.0 * npx * wts
        nbytes = npx * (c1_ - upc_ * 3 // 4 + c2_) * 2 + wts * 2  # block input (the upsampled channels at quarter size) + output + weights
        calls.append(("void c2f64_fused_kernel<%d, 10, %d>(C2f64Params)" % (nb_, 10 if nb_ == 2 else 20), flops, nbytes,
                      (lambda a=a: orig_c2f64(*a[:23], L.current_stream(dev)))))
    for a in btail_calls:  # (x, n, h, w, c, ldx, w3, b3, wt, bt, kind, nc, stride, y, a_total, a0, best_keys, dtype, opts, stream)
        npx, c_, kind, nc_ = a[1] * a[2] * a[3], a[4], a[10], a[11]
        cout = 64 if kind == 1 else nc_
        flops = 2.0 * npx * (9 * c_ * c_ + c_ * cout)
        nbytes = npx * c_ * 2 + npx * (4 if kind == 1 else nc_) * 4 + (9 * c_ * c_ + c_ * cout) * 2
        mt = 1 if (npx + 255) // 256 < torch.cuda.get_device_properties(dev).multi_processor_count else 2
        calls.append(("void conv_big_kernel<3, 1, 8, 1, %d, %d, %d>(BigParams)" % (mt, 4 if kind == 1 else (5 if c_ == 80 else 6), kind), flops, nbytes,
                      (lambda a=a: orig_btail(*a[:19], L.current_stream(dev)))))
    torch.cuda.synchronize(dev)
    fam = {}
    with torch.no_grad():
        for (name, flops, nbytes, replay) in calls:
            def body():
                for _ in range(reps):
                    replay()

            body()
            g = R.HipGraph()
            g.capture(body, device=dev)
            g.replay(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay(dev)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / reps
            d = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += ms
            d["flops"] += flops
            d["bytes"] += nbytes
    if torch.is_tensor(x) and model._stem_fusable(x, model._concat_placement()):
        # rows 0-1 run as one kernel that bypasses hip_conv2d (csrc/stem.hip: stem_conv_fused_kernel)
        n_, _, h_, w_ = x.shape
        with torch.no_grad(), R.static_buffers(pool):
            def body_f():
                for _ in range(reps):
                    model._fused_stem(x)
            body_f()
            g = R.HipGraph()
            g.capture(body_f, device=dev)
            g.replay(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay(dev)
            e1.record()
            torch.cuda.synchronize(dev)
        fl = 2.0 * n_ * ((h_ // 2) * (w_ // 2) * 16 * 27 + (h_ // 4) * (w_ // 4) * 32 * 144)
        by = n_ * 3 * h_ * w_ * es + n_ * (h_ // 4) * (w_ // 4) * 32 * es
        o_ = R.current_opts()
        fam["void stem_conv_fused_kernel<%d>(StemFusedParams)" % (4 if (o_ is not None and o_.stemf_waves == 4) else 8)] = dict(launches=1, ms=e0.elapsed_time(e1) / reps, flops=fl, bytes=float(by))
    conv_ms = sum(d["ms"] for d in fam.values())
    conv_flops = sum(d["flops"] for d in fam.values())
    conv_bytes = sum(d["bytes"] for d in fam.values())
    dom_name, dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
    peak = PEAK_BF16_TFLOPS if es == 2 else PEAK_F32_TFLOPS
    avg_s = dom["ms"] / dom["launches"] * 1e-3
    achieved_tf = dom["flops"] / dom["launches"] / avg_s / 1e12
    achieved_gbs = dom["bytes"] / dom["launches"] / avg_s / 1e9
    ai = dom["flops"] / dom["bytes"]
    bound = "mfma" if ai > peak * 1e12 / (PEAK_HBM_GBS * 1e9) else "hbm"
    roofline = {
        "kernel": dom_name,
        "bound": bound,
        "achieved": round(achieved_tf if bound == "mfma" else achieved_gbs, 2),
        "peak": peak if bound == "mfma" else PEAK_HBM_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": round((achieved_tf / peak) if bound == "mfma" else (achieved_gbs / PEAK_HBM_GBS), 4),
        "traffic": None,
        "launches_per_step": dom["launches"],
        "avg_launch_us": round(avg_s * 1e6, 2),
        "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
        "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
        "flop_per_byte": round(ai, 1),
        "achieved_tflops": round(achieved_tf, 2),
        "achieved_gbs": round(achieved_gbs, 1),
        "timing": f"HIP events around a hipGraph replay of {reps} back-to-back launches per layer, on the launch stream "
                  "(isolated kernel time; agrees with rocprofv3 AverageNs of `bench.py --serial`, while in the default "
                  "run the Detect branches overlap on side streams and rocprofv3 reports stretched durations)",
    }
    # HBM traffic per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see profiles/): only valid for
    # the configuration they were collected on
    for pf in sorted((ROOT / "profiles").glob("r*_pmc_hbm_summary.json"), reverse=True):  # the latest round that measured this kernel
        try:
            pmc = json.loads(pf.read_text())
            if pmc.get("config") == f"{args.model} bs={args.batch} {args.dtype}" and dom_name in pmc["kernels"]:
                roofline["traffic"] = round(pmc["kernels"][dom_name]["hbm_bytes_per_launch"])
                roofline["traffic_source"] = f"profiles/{pf.name} (rocprofv3 --pmc, separate passes)"
                break
        except (OSError, KeyError, ValueError):
            pass
    # What bounds the quoted mode: with several steps in flight the small-map launches of other steps hide under the chip-filling
    # ones, so the step is (nearly) the SUM of the launches that fill the chip by themselves - listed here, each against the tighter
    # of its two rooflines, with the vector-issue time of its instruction count (PMC INSTS_VALU per launch over 1024 SIMDs at the
    # measured 2.5 cycles per wave-instruction and 2.1 GHz; transcendental instructions cost 8.2, so this is a lower bound)
    valu = {}
    for pf in sorted((ROOT / "profiles").glob("r*_pmc_step_budget.txt"), reverse=True):
        try:
            for ln in pf.read_text().splitlines()[1:]:
                rest = ln[60:].split()  # (kernel name padded to 60 columns) calls INSTS_VALU INSTS_SALU ...
                if len(rest) > 2 and rest[0].isdigit():
                    valu.setdefault(ln[:60].strip(), float(rest[1]) / max(int(rest[0]), 1))
        except (OSError, ValueError, IndexError):
            pass
        break
    crit = []
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"] / kv[1]["launches"]):
        us = v["ms"] / v["launches"] * 1e3
        if us < 25.0:
            continue
        tf, gb = v["flops"] / (v["ms"] * 1e-3) / 1e12, v["bytes"] / (v["ms"] * 1e-3) / 1e9
        kn = k.replace("void ", "").split("(")[0]  # the PMC table strips "void " and the parameter list and cuts names at 60 columns
        vi = next((valu[n_] for n_ in valu if n_ and (kn == n_ or kn[:60].rstrip() == n_)), None)
        if vi is None:  # same kernel template, one instantiation in the table (its template list may be spelled with defaults)
            same = [n_ for n_ in valu if n_ and n_.split("<")[0] == kn.split("<")[0]]
            vi = valu[same[0]] if len(same) == 1 else None
        crit.append({"kernel": k, "launches": v["launches"], "avg_us": round(us, 1), "frac_mfma": round(tf / peak, 3),
                     "frac_hbm": round(gb / PEAK_HBM_GBS, 3), "frac_of_tighter_roofline": round(max(tf / peak, gb / PEAK_HBM_GBS), 3),
                     "valu_issue_us": None if vi is None else round(vi / 1024 * 2.5 / 2.1e3, 1)})
    kernels = {
        "critical_path": crit,
        "conv_ms_per_step": round(conv_ms, 4),
        "conv_tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 1),
        "conv_algorithmic_gbs": round(conv_bytes / (conv_ms * 1e-3) / 1e9, 1),
        "conv_hbm_floor_ms": round(conv_bytes / 6.0e12 * 1e3, 4),
        "conv_algorithmic_bytes": conv_bytes,
        "conv_launches_per_step": int(sum(d["launches"] for d in fam.values())),
        "families": {k: dict(launches=v["launches"], avg_us=round(v["ms"] / v["launches"] * 1e3, 2),
                             tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                             gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)) for k, v in sorted(fam.items())},
    }
    return roofline, kernels


def gpu_parity(args, dev, ppath, model, x0, results, pb):
    """GPU output vs the oracle's on the SAME batch (rank 0's first resident batch = procedural images 0 .. pb - 1), in the same run:
      f32  - the parity mode (`--dtype f32`: exact-f32 MFMA): one extra forward + NMS of an f32 copy of the model after the timed
             region; max |box| / |score| over every anchor of the head output, and the detections row by row (north_star: 1e-3);
      bf16 - the mode the throughput is quoted in: the detections the TIMED region itself produced for that batch (the static result
             of compiled copy 0) as a set against the oracle's (one-to-one same-class matches at IoU >= 0.9 / 0.5), plus the head
             output of one eager forward.
    The oracle's tensors come from the CPU-baseline child (`--parity-out`); nothing of oracle/ is imported here."""
    import numpy as np

    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import parity as PA
    from ultralytics_pro_amd.utils import procedural as P
    from ultralytics_pro_amd.utils.nms import non_max_suppression

    if not os.path.exists(ppath):
        return {"error": "the CPU leg produced no oracle output (cut at its wall-clock limit?)"}
    ref = torch.load(ppath)
    y_ref = ref["y"]
    ref_rows = PA.split_rows(ref["rows"].numpy(), ref["n"])
    out = {"images": int(y_ref.shape[0]), "oracle": f"oracle (CPU f32, fused eval, {ref['threads']} threads) on procedural images 0..{pb - 1}",
           "reference_detections": int(sum(ref["n"]))}
    if "rtdetr" in args.model:
        return _gpu_parity_rtdetr(args, dev, y_ref, model, x0, out, pb)
    with torch.no_grad():
        mf = DetectionModel(args.model + ".yaml")
        P.apply_procedural_weights(mf)
        mf = mf.to(dev).eval()
        mf.set_compute_dtype(torch.float32)
        x32 = P.synthetic_images(pb, first=0).to(dev)
        yf = mf(x32)[0]
        det = [d.cpu().numpy() for d in non_max_suppression(yf, 0.25, 0.7, max_det=300)]
        d = (yf.cpu() - y_ref).abs()
        eq, _, _ = PA.rows_identical(det, ref_rows, 1e-3)
        rq = PA.rows_equivalent(det, ref_rows, 1e-3, 0.25, 0.7)
        out["f32"] = {"max_box_abs_px": float(d[:, :4].max()), "max_score_abs": float(d[:, 4:].max()),
                      "detections": int(sum(len(r) for r in det)), "rows_equal": eq,
                      # rows whose presence a threshold decides within the tolerance (score within 1e-3 of conf, IoU within 1e-3 of
                      # iou_thres, or overlapping such a row) are counted and excused; every other row must have its partner
                      "rows": rq, "tolerance": 1e-3,
                      "within_tolerance": bool(d[:, :4].max() <= 1e-3 and d[:, 4:].max() <= 1e-3 and rq["equivalent"])}
        if args.model == "yolov8s":
            # the reference's OWN f32 output on this model moves by 2.2e-3 px between 8 and 1 CPU threads and sits 1.8e-3 .. 2.9e-3 px from
            # its float64 run (tools/ref_noise_floor.py; tests/test_oracle_golden.py): 1e-3 px is below its reproducibility there
            out["f32"]["box_tolerance_note"] = ("yolov8s: the reference's own f32 noise floor is 2.2e-3 - 2.9e-3 px (8 vs 1 threads, vs float64); "
                                                "the tests gate its boxes at 3e-3 px, scores at 1e-3")
            out["f32"]["within_reference_noise_floor"] = bool(d[:, :4].max() <= 3e-3 and d[:, 4:].max() <= 1e-3 and rq["equivalent"])
        del mf, yf
        if args.dtype == "bf16":
            mine = []
            for (o_, c_, _) in results:
                oc, cc = o_.cpu().numpy(), c_.cpu().tolist()
                mine += [oc[i, :int(cc[i])] for i in range(len(cc))]
            a9, a5 = PA.detection_agreement(mine, ref_rows, 0.9), PA.detection_agreement(mine, ref_rows, 0.5)
            det_ = model.model[-1]
            so_ = getattr(det_, "scores_out", True)
            if hasattr(det_, "scores_out"):
                det_.scores_out = True  # the head comparison reads the class rows
            try:
                yb = model(x0)[0].float().cpu()
            finally:
                if hasattr(det_, "scores_out"):
                    det_.scores_out = so_
            db = (yb - y_ref).abs()
            out["bf16"] = {"detections": a9["n_mine"], "recall_iou90": round(a9["recall"], 4), "precision_iou90": round(a9["precision"], 4),
                           "recall_iou50": round(a5["recall"], 4), "precision_iou50": round(a5["precision"], 4),
                           "matched_box_p99_px": round(a9["box_p99"], 4), "matched_box_max_px": round(a9["box_max"], 4),
                           "matched_score_max": round(a9["score_max"], 5),
                           "head_box_p99_px": float(np.quantile(db[:, :4].numpy().ravel()[::7], 0.99)), "head_box_max_px": float(db[:, :4].max()),
                           "head_score_max": float(db[:, 4:].max()),
                           "source": "detections: the timed region's own result for this batch (compiled copy 0); head: one eager forward"}
    return out


def _gpu_parity_rtdetr(args, dev, y_ref, model, x0, out, pb):
    """Config 5: the (B, 300, 4 + nc) decoder output against the oracle's as SETS of rows per image (the 300 queries are the top-300
    tokens by encoder score, head.py:2175: two implementations may order near-ties differently, and the bf16 mode may pick other
    tokens near the cut).  f32: the largest distance of an oracle row to its own partner (1e-3 = north_star's tolerance); bf16: the
    fraction of oracle rows reproduced within the reference's AMP tolerance (0.5 px of 640, utils/checks.py:780) and 0.01 per score."""
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    from ultralytics_pro_amd.utils import procedural as P

    def sets(y):
        worst, frac = 0.0, []
        for i in range(y.shape[0]):
            dist = (y[i][:, None, :] - y_ref[i][None, :, :]).abs().amax(2)
            worst = max(worst, float(dist.min(0).values.max()))
            db = (y[i][:, None, :4] - y_ref[i][None, :, :4]).abs().amax(2)
            j = db.argmin(0)
            sc = (y[i][j, 4:] - y_ref[i][:, 4:]).abs().amax(1)
            frac.append(float(((db.min(0).values <= 0.5 / 640) & (sc <= 0.01)).float().mean()))
        return worst, frac
    with torch.no_grad():
        mf = DetectionModel(args.model + ".yaml")
        P.apply_procedural_weights(mf)
        mf = mf.to(dev).eval()
        mf.set_compute_dtype(torch.float32)
        yf = mf(P.synthetic_images(pb, first=0).to(dev))[0].float().cpu()
        worst, _ = sets(yf)
        out["f32"] = {"worst_row_to_partner": worst, "tolerance": 1e-3, "within_tolerance": bool(worst <= 1e-3),
                      "compared": "decoder output rows as sets per image (normalised boxes, class scores)"}
        del mf, yf
        if args.dtype == "bf16":
            yb = model(x0)[0].float().cpu()
            _, frac = sets(yb)
            out["bf16"] = {"oracle_rows_reproduced_mean": round(sum(frac) / len(frac), 4), "oracle_rows_reproduced_min": round(min(frac), 4),
                           "within": "0.5 px of 640 on the box and 0.01 on every class score",
                           "note": "random-weight encoder scores are nearly flat: the bf16 mode selects other top-300 tokens near the cut, "
                                   "so agreement is bounded by the query-set overlap (tests/test_hip_e2e.py pins the backbone and the decoder separately)"}
    return out


def physical_cores() -> int:
    """Physical cores of this host (unique (physical id, core id) pairs of /proc/cpuinfo; logical count if unavailable)."""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline_child(args):
    """Child-process entry of the CPU baseline leg: never touches the GPU; the result file is rewritten after every line."""
    out = args.cpu_baseline_child
    res = run_cpu_train_baseline(args) if args.workload == "train" else run_cpu_baseline(args, progress=out, parity_out=args.parity_out)
    with open(out, "w") as f:
        json.dump(res, f)
    return 0


def run_cpu_baseline_bounded(args, limit_s: float = 150.0, parity_out: str | None = None):
    """Run the CPU baseline leg as a child process (`bench.py --cpu-baseline-child`) with a hard wall-clock limit.

    The leg times torch CPU convolutions at up to all physical cores of a host that bench.py does not own: on a busy or
    oversubscribed host a single forward can take minutes (it once stalled a whole default run), and an in-process forward
    cannot be interrupted.  The child keeps its result file current, so whatever was measured before the limit is reported
    (`"truncated": true`); the GPU numbers never wait for more than `limit_s`."""
    import subprocess
    import tempfile
    fd, out = tempfile.mkstemp(prefix="upa_cpu_baseline_", suffix=".json")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", out, "--workload", args.workload, "--model", args.model,
           "--batch", str(args.batch), "--imgsz", str(args.imgsz), "--cpu-threads", str(args.cpu_threads)]
    if parity_out:
        cmd += ["--parity-out", parity_out]
    env = dict(os.environ, OMP_WAIT_POLICY="passive", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    truncated = False
    try:
        subprocess.run(cmd, env=env, timeout=limit_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    except subprocess.TimeoutExpired:  # subprocess.run has killed the child
        truncated = True
    try:
        with open(out) as f:
            res = json.load(f)
    except (OSError, ValueError):
        res = None
    finally:
        try:
            os.unlink(out)
        except OSError:
            pass
    if res is None:
        return {"value": None, "unit": "images/s", "cores": 0, "kind": "port",
                "sample": f"CPU baseline leg produced nothing within {limit_s:.0f} s on this host (child process killed)"}
    if truncated:
        res["truncated"] = True
        res["sample"] += f"; the leg was cut at {limit_s:.0f} s wall clock, later (config, threads) lines are missing"
    return res


def _cpu_baseline_result(args, lines, logical, phys):
    head = [ln for ln in lines if ln["config"] == args.model and "forward_nms_img_s" in ln]
    if not head:
        return None
    top = max(head, key=lambda ln: ln["forward_nms_img_s"])
    return {"value": top["forward_nms_img_s"], "unit": "images/s", "cores": top["threads"], "kind": "port",
            "sample": f"oracle (torch CPU fp32, fused eval) {args.model} bs={args.batch} forward+NMS at {top['threads']} threads, best "
                      f"of {top['best_of']}; host: {logical} logical / {phys} physical cores; every (config, threads) line is in `lines`",
            "host_logical_cores": logical, "host_physical_cores": phys, "lines": lines}


def run_cpu_baseline(args, budget_s: float = 45.0, progress: str | None = None, parity_out: str | None = None):
    """the oracle (CPU restatement, validated bit for bit against the imported reference) on THIS
    host's cores - fused eval, fp32 - for C2 (yolov8n, 32 x 3 x 640 x 640) and C1 (yolov3-tiny, 8 x 3 x 640 x 640), with
    N = 8 threads (the reference's own cap NUM_THREADS = min(8, cpus - 1), utils/__init__.py:43), N = 32 and N = all physical
    cores; 1 warm-up, best of up to 3 (fewer when one pass is slow: the whole leg is bounded to ~`budget_s` seconds);
    forward and forward + NMS (conf 0.25, iou 0.7, max_det 300) as images/s and per-image ms in the reference's Profile
    format (validator.py:253-256).  `value` = the best forward+NMS rate of the headline config."""
    from oracle import nms as onms
    from oracle import tasks as ot
    from ultralytics_pro_amd.utils import procedural as P

    logical, phys = os.cpu_count() or 1, physical_cores()
    threads = [args.cpu_threads] if args.cpu_threads else sorted({min(8, logical), min(32, logical), phys})
    configs = [(args.model, args.batch)] + ([("yolov3-tiny", 8)] if args.model == "yolov8n" else [])
    lines, t_start = [], time.perf_counter()
    for name, b in configs:
        m = ot.DetectionModel(name + ".yaml")
        P.apply_procedural_weights(m)
        m.fuse()
        x = P.synthetic_images(b)
        per_img_best = None
        for nthr in threads:
            torch.set_num_threads(nthr)
            with torch.no_grad():
                t0 = time.perf_counter()
                m(x[:2])  # warm-up and oversubscription probe (torch CPU convs collapse when threads >> useful cores)
                probe = (time.perf_counter() - t0) / 2
                left = budget_s - (time.perf_counter() - t_start)
                if (per_img_best is not None and probe > 6 * per_img_best) or probe * b > left:
                    lines.append({"config": name, "batch": b, "threads": nthr, "skipped": f"probe {probe * 1e3:.0f} ms/image at "
                                  f"bs 2: slower than fewer threads or over the time budget"})
                    continue
                reps = max(1, min(3, int(left / 3 / max(probe * b, 1e-3))))
                best_f = best_n = 1e30
                for _ in range(reps):
                    t0 = time.perf_counter()
                    y = m(x)[0]
                    t1 = time.perf_counter()
                    post = (lambda yy: onms.rtdetr_postprocess(yy, 0.25)) if "rtdetr" in name else (lambda yy: onms.non_max_suppression(yy, 0.25, 0.7, max_det=300))
                    post(y)
                    t2 = time.perf_counter()
                    best_f, best_n = min(best_f, t1 - t0), min(best_n, t2 - t1)
                    if parity_out and name == args.model and not os.path.exists(parity_out):
                        # the oracle's answer on the GPU's first resident batch (same procedural images and weights): the parent compares
                        det = post(y)
                        torch.save({"y": y, "rows": torch.cat(det, 0), "n": [int(d.shape[0]) for d in det], "threads": nthr,
                                    "first_image": 0, "batch": b}, parity_out + ".tmp")
                        os.replace(parity_out + ".tmp", parity_out)
            per_img_best = min(per_img_best or 1e30, best_f / b)
            lines.append({"config": name, "batch": b, "threads": nthr, "best_of": reps,
                          "forward_img_s": round(b / best_f, 2), "forward_nms_img_s": round(b / (best_f + best_n), 2),
                          "speed": "Speed: %.1fms preprocess, %.1fms inference, %.1fms loss, %.1fms postprocess per image" % (
                              0.0, best_f / b * 1e3, 0.0, best_n / b * 1e3)})
            if progress:  # keep the parent's view current: it may have to kill this process at its wall-clock limit
                part = _cpu_baseline_result(args, lines, logical, phys)
                if part is not None:
                    with open(progress + ".tmp", "w") as f:
                        json.dump(part, f)
                    os.replace(progress + ".tmp", progress)
    res = _cpu_baseline_result(args, lines, logical, phys)
    if res is None:
        raise RuntimeError("CPU baseline: no line of the headline config was measured")
    return res


if __name__ == "__main__":
    main()
