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

from tools.bench_legs.common import GFLOP_OTHER, GFLOP_PER_IMG, PEAK_BF16_TFLOPS, PEAK_F32_TFLOPS, PEAK_HBM_GBS  # noqa: E402,F401
from tools.bench_legs.cpu_baseline import (cpu_baseline_child, physical_cores, run_cpu_baseline, run_cpu_baseline_bounded,  # noqa: E402,F401
                                           run_cpu_train_baseline)
from tools.bench_legs.parity import gpu_parity  # noqa: E402
from tools.bench_legs.profile import kernel_profile, step_roofline, wgrad_profile  # noqa: E402


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


def _gc_off() -> bool:
    """The timed region enqueues its steps from this thread: a cyclic-GC pass of the interpreter in the middle of a 10 ms region (the driver's
    --steps 20 form) starves the lanes for tens of milliseconds - two of ~70 short runs in round 6 came back at 8.4 k and 15.9 k images/s
    instead of 59-61 k.  The collector is switched off from the first warm-up step to the end of the timed region; it is NOT run first: a
    collection (~0.1 s of host time with torch loaded) in front of the warm-up is an idle gap in which the GPU's clocks drop, and the 5 + 20
    steps of the driver's form are too short to ramp them back up (measured: 51 k against 60 k images/s)."""
    import gc
    was = gc.isenabled()
    gc.disable()
    return was


def _gc_on(was: bool) -> None:
    import gc
    if was:
        gc.enable()


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
        gc_was = _gc_off()
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
        _gc_on(gc_was)
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
    mode = {} if args.no_mode_dispatch else {k: v for k, v in {"c2f": 4, "conv_ws3": 1, "c2f_stream_rows": -1, "detect_stream": 2, "conv_big": 2}.items() if k not in given}
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
    gc_was = _gc_off()
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
    _gc_on(gc_was)
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
    gc_was = _gc_off()
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
    _gc_on(gc_was)
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


if __name__ == "__main__":
    main()
