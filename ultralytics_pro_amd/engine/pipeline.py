"""Throughput executor: several compiled copies of the inference step kept in flight on separate HIP streams.

One hipGraph replay of the step ends in a latency-bound tail (Detect decode + NMS: a few dozen workgroups on 256 CUs).
Replaying consecutive batches round-robin on `in_flight` streams - each with its own graph and static buffers - lets the
tail of batch k overlap the convolutions of batch k+1, the way a serving loop keeps several requests in flight.  How well
streams overlap depends on how the HIP runtime maps them onto hardware queues (GPU_MAX_HW_QUEUES, creation order), so
`autotune` measures a few (in_flight, micro_batches) combinations for a handful of steps and keeps the fastest.
"""

from __future__ import annotations

import contextlib
import os
import sys
import time

import torch


class PipelinedRunner:
    """`example` may be ONE resident input batch or a LIST of distinct resident batches: every batch gets its own compiled
    copy of the step (graph + static buffers) and the copies are replayed round-robin on `in_flight` lane streams.  With
    >= 8 distinct 78.6 MB batches the inputs (629 MB) exceed the 256 MB Infinity Cache, so the input read of a step is a
    real HBM read and not a replay of one cache-resident batch."""

    def __init__(self, model, example, post=None, micro_batches: int = 2, in_flight: int = 2, priority: int = 0,
                 sub_priority: int = 0, linear: bool = False, mode_dispatch: bool = True):
        examples = list(example) if isinstance(example, (list, tuple)) else [example]
        example = examples[0]
        self.model, self.example, self.post = model, example, post
        self.micro_batches, self.in_flight, self.priority = micro_batches, max(1, in_flight), priority
        self.device = example.device
        # linear: every copy is ONE chain of launches (no forks inside the graph: Detect branches stay on the capture
        # stream) - the copies, one stream each, are the only concurrency, which maps cleanly onto the hardware queues
        self.linear = bool(linear)
        det = model.model[-1]
        saved = getattr(det, "concurrent", None)
        if self.linear and saved is not None:
            det.concurrent = False
        # Dispatch by execution mode.  With several steps in flight the chip is shared by kernels of different steps and what counts is
        # the LDS / register time a launch holds, not its latency alone:
        #   * the whole-block kernel of the 40 x 40 C2f blocks (csrc/c2f64.hip) spends 1.74x the useful MFMAs on its tile rings: 2 % less
        #     throughput in flight (51.6 k vs 50.5 k img/s, same box), 4 % faster with one step at a time (1.012 vs 1.054 ms);
        #   * the persistent 3x3 kernel (csrc/conv_ws3.hip) is 12 % faster than conv_big launch for launch, but its resident workgroups
        #     keep their CUs until the launch ends: -0.7 % in flight (51.6 k vs 52.0 k), neutral to +0.2 % one step at a time.
        #   * the line-buffer C2f kernel (csrc/c2f_stream.hip) pays 11 steps of pipeline fill per workgroup: whole-height strips (128 workgroups
        #     for the 80 x 80 maps at batch 32: half the CUs, the other steps' kernels take the rest) hold 15 % less CU time than the two
        #     parts per strip that fill the chip with one round: 56.0 k vs 55.4 k img/s in flight (same box), 90 vs 53 us one step at a time.
        #   * the line-buffer form of the 80 x 80 Detect level (csrc/detect_stream.hip) is SLOWER launch for launch (one step at a time 0.878 vs
        #     0.799 ms: 192 long-lived workgroups, two waves per SIMD) but holds a quarter less CU time than the tile form's two chip-filling
        #     launches and moves 130 MB less: 59.4 k vs 58.1 k images/s in flight, same box (round 6).
        #   * conv_big for every shape it can run (conv_big = 2) instead of the size rule, which sends the smallest stride-2 / pointwise layers to
        #     conv_igemm (faster alone, but its workgroups re-read their weights through L1 and yolov8n's model.19 moved 2.2x its algorithmic
        #     bytes): 62.27 k vs 61.74 k images/s in flight (+0.85 %; yolov8s +1.6 %; the validate path, tiny, BoT3, rtdetr within noise), 0.5 % slower
        #     one step at a time (round 6).
        # So: in flight > 1 -> upa_opts.c2f = 4, conv_ws3 = 1, c2f_stream_rows = -1, detect_stream = 2 and conv_big = 2 for the compiled copies,
        # unless the caller's options already set them.
        from . import runtime as R
        cur = R.current_opts()
        mode = {}
        if mode_dispatch and self.in_flight > 1 and getattr(model, "opts", None) is None:
            if cur is None or cur.c2f == 0:
                mode["c2f"] = 4
            if cur is None or cur.conv_ws3 == 0:
                mode["conv_ws3"] = 1
            if cur is None or cur.c2f_stream_rows == 0:
                mode["c2f_stream_rows"] = -1
            if cur is None or cur.detect_stream == 0:
                mode["detect_stream"] = 2
            if cur is None or cur.conv_big == 0:
                mode["conv_big"] = 2
        self.throughput_opts = dict(mode)
        try:
            with torch.no_grad(), (R.use_opts(**mode) if mode else contextlib.nullcontext()):
                ncopies = max(self.in_flight, len(examples))
                self.runs = [model.compile(examples[j % len(examples)], post=post, micro_batches=micro_batches,
                                           stream_priority=sub_priority) for j in range(ncopies)]
        finally:
            if self.linear and saved is not None:
                det.concurrent = saved
        # lane streams; `priority` (0 normal, -1 high) selects the runtime's queue set for them
        self.lanes = [torch.cuda.Stream(device=self.device, priority=priority) for _ in range(self.in_flight)]
        self.i = 0

    def step(self):
        """Enqueue one batch; returns the (static) result object of the copy that ran it - valid after a synchronize."""
        j = self.i % len(self.runs)
        self.i += 1
        with torch.cuda.stream(self.lanes[j % self.in_flight]):
            return self.runs[j]()

    def results(self):
        return [r.result for r in self.runs]

    def measure(self, steps: int = 20, warmup: int = 4) -> float:
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize(self.device)
        return (time.perf_counter() - t0) / steps


def _trace(msg):
    if os.environ.get("UPA_BENCH_TRACE"):
        print(f"[pipeline] {msg}", file=sys.stderr, flush=True)


def autotune(model, example, post=None, candidates=((4, 1, 0, 1), (2, 2, -1, 0), (3, 2, 0, 0), (1, 2, 0, 0)), steps: int = 20,
             mode_dispatch: bool = True):
    """Try (in_flight, micro_batches, lane priority, linear) candidates; returns (best runner, {candidate: s/step})."""
    best, best_t, table = None, float("inf"), {}
    for cand in candidates:
        in_flight, mb = cand[0], cand[1]
        prio = cand[2] if len(cand) > 2 else 0
        linear = bool(cand[3]) if len(cand) > 3 else False
        if (example[0] if isinstance(example, (list, tuple)) else example).shape[0] % mb:
            continue
        _trace(f"candidate {cand}: compile")
        r = PipelinedRunner(model, example, post, micro_batches=mb, in_flight=in_flight, priority=prio, linear=linear,
                            mode_dispatch=mode_dispatch)
        _trace(f"candidate {cand}: measure")
        # best of three short measurements: a single 20-step sample is sometimes 5x off (a one-time runtime hiccup inside the
        # window: 4.08 ms instead of 0.71 in one of eight otherwise identical processes), which made the tuner pick a slower
        # configuration and the whole run land 8 % low
        t = min(r.measure(steps) for _ in range(3))
        _trace(f"candidate {cand}: {t * 1e3:.3f} ms")
        table[(in_flight, mb, prio, int(linear))] = t
        if t < best_t:
            best, best_t = r, t
    return best, table
