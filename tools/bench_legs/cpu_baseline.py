"""bench.py's `cpu_baseline` leg: the oracle (CPU restatement of the reference path) timed on this host's cores, in a CHILD process with a hard
wall-clock limit.  The only leg that imports oracle/ - as the thing being timed beside the GPU number, never as part of the product path."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import BENCH_PY


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


def _oracle_rtdetr_encoder_side(m, x):
    """(CPU child, checker side) One more oracle forward of yolov3-rtdetr with the encoder-side tensors of `_get_decoder_input` kept
    (head.py:2143-2200): class probabilities and encoder boxes of EVERY token and the top-300 token indices - what the parent needs to
    compare the bf16 mode in front of the (chaotic) query selection and, with those indices injected, behind it."""
    head = m.model[-1]
    kept = {}
    h1 = head.enc_output.register_forward_hook(lambda mod, i, o: kept.__setitem__("features", o.detach()))
    h2 = head.enc_score_head.register_forward_hook(lambda mod, i, o: kept.__setitem__("scores", o.detach()))
    try:
        m(x)
    finally:
        h1.remove()
        h2.remove()
    sz = [int(x.shape[2]) // s for s in (8, 16, 32)]
    anchors, valid = head._generate_anchors([[s, s] for s in sz])
    return {"enc_prob": kept["scores"].sigmoid(), "enc_box": (head.enc_bbox_head(kept["features"]) + anchors).sigmoid(),
            "enc_valid": valid.view(-1), "topk": torch.topk(kept["scores"].max(-1).values, head.num_queries, dim=1).indices}


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
    cmd = [sys.executable, str(BENCH_PY), "--cpu-baseline-child", out, "--workload", args.workload, "--model", args.model,
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
                        rec = {"y": y, "rows": torch.cat(det, 0), "n": [int(d.shape[0]) for d in det], "threads": nthr,
                               "first_image": 0, "batch": b}
                        if "rtdetr" in name:
                            rec.update(_oracle_rtdetr_encoder_side(m, x))
                        torch.save(rec, parity_out + ".tmp")
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
