"""Detection heads with the reference's operator API (ultralytics/nn/modules/head.py), on HIP kernels."""

from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import _lib as L
from ...engine import runtime as R
from .block import DFL
from .conv import Conv, _HipConvMixin, hip_conv2d, version_key

__all__ = ("Detect",)


def _magic_exact(xmax: int, d: int) -> bool:
    """csrc/detect_epi.h upa_magic_exact: umulhi(x, floor(2^32 / d) + 1) == x // d for every 0 <= x <= xmax."""
    if xmax < 0 or d <= 0 or xmax > 0xFFFFFFFF:
        return False
    e = ((1 << 32) // d + 1) * d - (1 << 32)
    return xmax * e < (1 << 32)


class Detect(nn.Module, _HipConvMixin):
    """YOLO Detect head (head.py:28-191), legacy v3/v5/v8 class branch (:98-100).

    eval forward returns `(y, x)` like the reference: y = (B, 4+nc, A) float32 decoded boxes (xywh, pixels) and class
    probabilities, x = the per-level raw maps (NHWC views, logical (B, 4*reg_max+nc, H, W)).
    Per level the box and class branches write the two channel slices of one (B,H,W,144) buffer (the reference's
    `torch.cat((cv2(x), cv3(x)), 1)`, head.py:122) and one `upa_detect_decode` launch performs
    DFL + dist2bbox + stride scaling + sigmoid (head.py:151-169).
    """

    dynamic = False
    export = False
    format = None
    end2end = False
    max_det = 300
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)
    legacy = False
    xyxy = False

    def __init__(self, nc: int = 80, ch: tuple = ()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        if not self.legacy:
            raise L.UpaError("Detect(legacy=False) (DWConv class branch, v11+) is outside the hot-path scope (SURVEY §2)")
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()

    # ---- decode fused into the last 1x1 conv of each branch (bf16 perf mode) ----------------------------------------------
    # `upa_detect_tail`: the branch's final nn.Conv2d and its half of Detect._inference (DFL + dist2bbox + stride, or the
    # class sigmoid; head.py:151-169) are ONE kernel - the (B, H, W, 4*reg_max + nc) logits stay in MFMA accumulators.
    # `keep_raw` decides whether they are ALSO written out as the second return value (the reference's `(y, x)`,
    # head.py:126; the predict / validate hot path only reads y).  f32 parity mode keeps conv + upa_detect_decode.
    fuse_decode = True
    keep_raw = True

    def _plan(self):
        return self.__dict__.setdefault("_plans", {})

    def begin(self, n: int, level_hw, dtype, device) -> None:
        """Allocate the decoded output for a forward whose Detect inputs will have the spatial sizes `level_hw` (known
        statically from the input size) so that every level's branches can decode straight into their anchor range."""
        a0, tot = [], 0
        for (h, w) in level_hw:
            a0.append(tot)
            tot += int(h) * int(w)
        fused = bool(self.fuse_decode and dtype == torch.bfloat16 and self.reg_max == 16 and self.nc <= 128
                     and all(h * w >= 2 and w >= 2 for h, w in level_hw)
                     # the fused tails split a flat pixel index into (image, anchor) with a multiply-high: only exact up to a
                     # bound (1280x1280 bs 8 is past it) - larger shapes take conv + upa_detect_decode
                     and all(_magic_exact(n * h * w - 1, h * w) and _magic_exact(h * w - 1, w) for h, w in level_hw))
        y = R.alloc_plain((n, 4 + self.nc, tot), torch.float32, device, key=(id(self), "y"))
        plan = dict(y=y, a0=a0, a_total=tot, fused=fused, hw=[(int(h), int(w)) for h, w in level_hw], n=int(n), decoded=set(),
                    hot=None, hot_levels=set())
        if fused and self.nms_keys:
            # NMS prefilter: the class launches also write every anchor's best-class NMS key (upa_detect_branch_tail, or
            # upa_detect_tail for class branches outside that form / with raw maps kept)
            plan["hot"] = R.alloc_plain((n, tot), torch.int64, device, key=(id(self), "best_keys"))  # u64 bit patterns
        self._plan()[R.current_tag()] = plan

    # `upa_detect_branch_tail`: with no raw output wanted, the branch's SECOND 3x3 conv joins that launch too (its SiLU'd
    # accumulators are the MFMA operand of the 1x1 conv), so a branch is two launches: conv3x3, [conv3x3 + 1x1 + decode].
    fuse_branch = True
    # NMS prefilter (off by default): the fused class tails also write the NMS sort key of every anchor's best class, a dense
    # (B, A) array that `utils.nms` compacts instead of re-reading the (B, nc, A) scores (upa_nms_batched_hot; single-label NMS)
    nms_keys = False
    # `scores_out = False` (only with `nms_keys` and without `keep_raw`): the class rows of the decoded output are NOT written
    # (`upa_opts.keys_only`) - single-label `non_max_suppression` takes every anchor's best class and score from the keys and the
    # boxes from rows 0-3, so for predict the (B, nc, A) scores (86 MB per batch-32 step of yolov8n) and all but one sigmoid per
    # anchor are dead work.  The detections are bit-identical; rows 4.. of the returned tensor are then UNDEFINED (a consumer
    # that reads them - multi_label NMS, validation - must leave this at True).
    scores_out = True

    def _keys_only(self) -> bool:
        return bool(self.nms_keys and not self.scores_out and not self.keep_raw)

    def _tail(self, seq: nn.Sequential, x: torch.Tensor, raw: torch.Tensor | None, kind: int, i: int, plan) -> None:
        """conv3x3 -> conv3x3 -> [1x1 + decode] of one branch (kind 1 = box, 2 = class) of level i."""
        t = seq[0](x)
        if raw is None and self.fuse_branch and self._branch_tail(t, seq[1], seq[2], kind, i, plan):
            return
        self._tail_call(seq[1](t), seq[2], raw, kind, i, plan)

    def _branch_tail_args(self, t: torch.Tensor, mid, conv: nn.Conv2d, kind: int):
        """(view of t, packed 3x3, packed 1x1 tail, its bias) for `upa_detect_branch_tail`, or None outside the fused form."""
        c = mid.conv.in_channels
        cp = 64 if kind == 1 else (80 if c == 80 else 96)  # padded channel count of the fused form (upa_detect_branch_tail)
        if not (isinstance(mid, Conv) and isinstance(mid.act, nn.SiLU) and mid.conv.kernel_size == (3, 3) and mid.conv.stride == (1, 1)
                and mid.conv.padding == (1, 1) and mid.conv.groups == 1 and mid.conv.out_channels == c == conv.in_channels
                and t.dtype == torch.bfloat16 and conv.kernel_size == (1, 1)
                and (c == 64 if kind == 1 else (c <= 96 and c % 8 == 0 and self.nc <= 96))):
            return None
        pk3 = self._packed(mid.conv, mid.bn, t.device, t.dtype, False, pad_cout=cp)
        cache = self.__dict__.setdefault("_pk_cache", {})
        key = (id(conv), str(t.device), "tail", cp)
        ver = version_key(conv.weight, conv.bias)
        hit = cache.get(key)
        if hit is None or hit[0] != ver:
            w = torch.zeros(cp, cp)
            b = torch.zeros(cp)
            w[:conv.out_channels, :c] = conv.weight.detach().float().reshape(conv.out_channels, c).cpu()
            if conv.bias is not None:
                b[:conv.out_channels] = conv.bias.detach().float().cpu()
            host = torch.empty(L.lib().upa_tail_packed_weight_bytes(cp, cp), dtype=torch.uint8)
            L.check(L.lib().upa_pack_tail_weight(w.data_ptr(), cp, cp, host.data_ptr()), "pack_tail_weight")
            hit = (ver, (host.to(t.device), b.to(t.device)))
            cache[key] = hit
        wt, bt = hit[1]
        return R.view_of(t), pk3, wt, bt

    def _branch_tail(self, t: torch.Tensor, mid, conv: nn.Conv2d, kind: int, i: int, plan) -> bool:
        """[conv3x3 + SiLU + 1x1 + decode] in one launch; False when the branch is outside the fused form."""
        args = self._branch_tail_args(t, mid, conv, kind)
        if args is None:
            return False
        vt, pk3, wt, bt = args
        hot = plan.get("hot") if kind == 2 else None
        rc = L.lib().upa_detect_branch_tail(vt.ptr, vt.n, vt.h, vt.w, vt.c, vt.ld, pk3.w.data_ptr(), pk3.bias.data_ptr(),
                                            wt.data_ptr(), bt.data_ptr(), kind, self.nc, float(self.stride[i]),
                                            plan["y"].data_ptr(), plan["a_total"], plan["a0"][i],
                                            hot.data_ptr() if hot is not None else None, vt.dtype, R.opts_ptr(),
                                            L.current_stream(t.device))
        if rc == L.UPA_EUNSUPPORTED:
            return False
        L.check(rc, "detect_branch_tail")
        if hot is not None:
            plan["hot_levels"].add(i)
        return True

    # ---- one level, both branches, all six convolutions + the decode in ONE line-buffer launch (csrc/detect_stream.hip) ----------------
    # yolov8n's 80 x 80 level (64 input channels; 64-channel box branch, 80-channel class branch): the stacked first conv + the grouped
    # branch tails write and re-read a 144-channel intermediate and reload every weight slab per tile; the streaming form keeps t1 / t2 as
    # a few rows in LDS and the weights in registers.  Slower launch for launch (0.878 vs 0.799 ms one step at a time: two waves per SIMD, every
    # role bound by its own in-order issue, profiles/r06_detect_stream.txt) but 192 workgroups that hold a quarter less CU time: +2.4 % with four
    # steps in flight.  So it runs on request: `upa_opts.detect_stream = 2`, which `engine/pipeline.py` sets for copies in flight.
    level_stream = True

    def _level_stream(self, i: int, x: torch.Tensor, plan) -> bool:
        """Level i through `upa_detect_level_stream`; False (nothing launched) outside its form."""
        o_ = R.current_opts()
        if not self.level_stream or o_ is None or o_.detect_stream != 2 or x.dtype != torch.bfloat16 or self.reg_max != 16:
            return False
        b, c = self.cv2[i], self.cv3[i]
        ok3 = lambda m, ci, co: (isinstance(m, Conv) and not m.training and isinstance(m.act, nn.SiLU) and hasattr(m, "bn")  # noqa: E731
                                 and m.conv.kernel_size == (3, 3) and m.conv.stride == (1, 1) and m.conv.padding == (1, 1)
                                 and m.conv.groups == 1 and m.conv.in_channels == ci and m.conv.out_channels == co)
        ok1 = lambda m, ci, co: (isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)  # noqa: E731
                                 and m.groups == 1 and m.in_channels == ci and m.out_channels == co)
        cin = int(x.shape[1])
        if not (cin == 64 and self.nc <= 80 and ok3(b[0], 64, 64) and ok3(b[1], 64, 64) and ok1(b[2], 64, 64)
                and ok3(c[0], 64, 80) and ok3(c[1], 80, 80) and ok1(c[2], 80, self.nc)):
            return False
        dev, dt = x.device, x.dtype
        br = []
        for seq, cw, ct in ((b, 64, 64), (c, 80, 80)):
            pk1 = seq[0]._packed(seq[0].conv, seq[0].bn, dev, dt, False)
            pk2 = seq[1]._packed(seq[1].conv, seq[1].bn, dev, dt, False)
            pkt = self._packed(seq[2], None, dev, dt, False, pad_cout=ct)
            br.append(L.DetectBranch(cw, 0, pk1.w.data_ptr(), pk1.bias.data_ptr(), pk2.w.data_ptr(), pk2.bias.data_ptr(), pkt.w.data_ptr(),
                                     pkt.bias.data_ptr()))
        import ctypes as C
        vx = R.view_of(x)
        hot = plan.get("hot")
        rc = L.lib().upa_detect_level_stream(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, C.byref(br[0]), C.byref(br[1]), self.nc,
                                             float(self.stride[i]), plan["y"].data_ptr(), plan["a_total"], plan["a0"][i],
                                             hot.data_ptr() if hot is not None else None, L.dtype_code(dt), R.opts_ptr(),
                                             L.current_stream(dev))
        if rc == L.UPA_EUNSUPPORTED:
            return False
        L.check(rc, "detect_level_stream")
        if hot is not None:
            plan["hot_levels"].add(i)
        return True

    # ---- several levels per launch (linear graphs: the levels run one after the other on one stream anyway) ---------------------
    # The 40 x 40 and 20 x 20 levels' launches are 400 and 100 workgroups at batch 32 - a fraction of a round each, mostly launch
    # ramp and halo latency.  `upa_conv2d_bias_act_group` / `upa_detect_branch_tail_group` put problems that use the same kernel
    # instantiation into ONE grid, so the smaller level rides inside the larger one's partial round: 4 launches fewer per step.  (All
    # three levels in one grid - the 80 x 80 level on the 128-pixel variant - measured slower: upa_opts.no_group = 2.)
    group_levels = True
    stack_first = True  # the first level's two first convs as one 144-channel convolution (see _levels_grouped)

    def _levels_grouped(self, idx, xs, plan, allow_stack: bool = True) -> bool:
        """Both branches of the levels `idx` (inputs `xs`, NHWC) through the group entry points; False (nothing launched) when a
        level is outside the fused forms."""
        import ctypes as C
        lib = L.lib()
        dev = xs[0].device
        stream = L.current_stream(dev)
        todo = []
        for kind, seqs in ((1, self.cv2), (2, self.cv3)):  # check everything before the first launch
            for i in idx:
                c0 = seqs[i][0]
                if not (isinstance(c0, Conv) and not c0.training and c0.conv.kernel_size == (3, 3) and c0.conv.stride == (1, 1)
                        and c0.conv.padding == (1, 1) and c0.conv.groups == 1 and isinstance(c0.act, nn.SiLU) and hasattr(c0, "bn")):
                    return False
            todo.append((kind, seqs))
        # stage 1: the first convs of all branches in ONE group call.  Order = which neighbours may share a grid: the two branches of
        # the first (largest) level read the same input and use the same workgroup size, then each kind's remaining levels
        # The two first convs of the first (largest) level read the same input: stacked along the output channels they are ONE 64 + 80 =
        # 144-channel convolution (conv_big's nine-tile variant) - the halo of a tile is fetched once instead of once per branch, the box
        # branch multiplies no padded fifth tile, half the workgroups - and each branch's second half reads its channel slice of the
        # 144-channel tensor.  Same products in the same order per output channel as conv_big's two-problem grid: bit-identical to it -
        # NOT to the level-by-level walks under library-default options, whose 64 -> 64 box conv takes conv_ws3 (another f32 summation order:
        # last-bit differences, tests/test_hip_e2e.py::test_e2e_grouped_detect_levels_equal_level_by_level runs under conv_ws3 = 1 for that
        # reason; a default-options tolerance test sits next to it).  Measured (yolov8n bs 32, round 5, same
        # box, upa_opts.no_stack_first = 1 | 0): the stacked launch takes what the two-problem grid took (65.9 vs 66.1 us with four steps
        # in flight, 58.46 k images/s either way; the serial step 0.810 -> 0.808 ms): the first convs are bound by their per-tap matrix /
        # LDS work, not by the second halo fetch.  Kept: one launch and one tensor fewer, never slower.
        o_ = R.current_opts()
        i0 = idx[0]
        b0, c0_ = self.cv2[i0][0], self.cv3[i0][0]
        stacked = (allow_stack and self.stack_first and (o_ is None or not o_.no_stack_first) and xs[0].dtype == torch.bfloat16
                   and b0.conv.out_channels == 64 and c0_.conv.out_channels == 80 and b0.conv.in_channels == c0_.conv.in_channels
                   and b0.conv.in_channels >= 64)
        order = ([("stack", None, i0)] if stacked else [(kind, seqs, i0) for kind, seqs in todo]) + \
                [(kind, seqs, i) for kind, seqs in todo for i in idx[1:]]
        probs = (L.ConvProblem * len(order))()
        mid_of = {}
        for j, (kind, seqs, i) in enumerate(order):
            x = xs[idx.index(i)]
            vx = R.view_of(x)
            if kind == "stack":
                pk = self._packed_stack([(b0.conv, b0.bn), (c0_.conv, c0_.bn)], dev, x.dtype)
                t = R.alloc_nhwc(vx.n, pk.cout, vx.h, vx.w, x.dtype, dev, key=(id(b0), "y144"))
                mid_of[(1, i)], mid_of[(2, i)] = t[:, :64], t[:, 64:]
            else:
                c0 = seqs[i][0]
                pk = c0._packed(c0.conv, c0.bn, dev, x.dtype, False)
                t = R.alloc_nhwc(vx.n, pk.cout, vx.h, vx.w, x.dtype, dev, key=(id(c0), "y"))
                mid_of[(kind, i)] = t
            vy = R.view_of(t)
            probs[j] = L.ConvProblem(vx.ptr, vx.n, vx.h, vx.w, vx.c, vx.ld, pk.w.data_ptr(), pk.bias.data_ptr(), vy.ptr, pk.cout, vy.ld, None, 0)
        L.check(lib.upa_conv2d_bias_act_group(C.cast(probs, C.c_void_p), len(order), 3, 1, 1, L.ACT_SILU, L.dtype_code(xs[0].dtype),
                                              R.opts_ptr(), stream), "conv2d_group")
        # stage 2: the branch tails of both kinds in one call (box and class problems share grids where their workgroup sizes agree)
        args = {kind: [self._branch_tail_args(mid_of[(kind, i)], seqs[i][1], seqs[i][2], kind) for i in idx] for kind, seqs in todo}
        hot = plan.get("hot")
        rc = L.UPA_EUNSUPPORTED
        if all(a is not None for k in args for a in args[k]):
            lvs = {}
            for kind in (1, 2):
                lv = (L.BranchLevel * len(idx))()
                for j, (i, (vt, pk3, wt, bt)) in enumerate(zip(idx, args[kind])):
                    lv[j] = L.BranchLevel(vt.ptr, vt.n, vt.h, vt.w, vt.c, vt.ld, pk3.w.data_ptr(), pk3.bias.data_ptr(), wt.data_ptr(),
                                          bt.data_ptr(), float(self.stride[i]), plan["a0"][i])
                lvs[kind] = lv
            rc = lib.upa_detect_head_tails(C.cast(lvs[1], C.c_void_p), C.cast(lvs[2], C.c_void_p), len(idx), self.nc, plan["y"].data_ptr(),
                                           plan["a_total"], hot.data_ptr() if hot is not None else None, L.dtype_code(xs[0].dtype),
                                           R.opts_ptr(), stream)
        if rc == L.UPA_EUNSUPPORTED:  # a branch outside the branch-tail form: the second halves level by level
            for kind, seqs in todo:
                for i in idx:
                    t = mid_of[(kind, i)]
                    if not self._branch_tail(t, seqs[i][1], seqs[i][2], kind, i, plan):
                        self._tail_call(seqs[i][1](t), seqs[i][2], None, kind, i, plan)
        else:
            L.check(rc, "detect_head_tails")
            if hot is not None:
                plan["hot_levels"].update(idx)
        return True

    def _tail_call(self, t: torch.Tensor, conv: nn.Conv2d, raw, kind: int, i: int, plan) -> None:
        """The fused launch itself: t = the branch's second 3x3 output, conv = its final nn.Conv2d."""
        cout = 4 * self.reg_max if kind == 1 else self._ncp(t.dtype)
        pk = self._packed(conv, None, t.device, t.dtype, False, pad_cout=cout)
        vt = R.view_of(t)
        rp, rld = (None, 0)
        if raw is not None:
            vr = R.view_of(raw)
            rp, rld = vr.ptr, vr.ld
        hot = plan.get("hot") if kind == 2 else None
        L.check(L.lib().upa_detect_tail(vt.ptr, vt.n, vt.h, vt.w, vt.c, vt.ld, pk.w.data_ptr(), pk.bias.data_ptr(), cout, kind,
                                        self.nc, float(self.stride[i]), plan["y"].data_ptr(), plan["a_total"], plan["a0"][i],
                                        rp, rld, hot.data_ptr() if hot is not None else None, vt.dtype, R.opts_ptr(),
                                        L.current_stream(t.device)), "detect_tail")
        if hot is not None:
            plan["hot_levels"].add(i)

    def _branch(self, seq: nn.Sequential, x: torch.Tensor, out: torch.Tensor) -> None:
        t = seq[1](seq[0](x))
        # `out` may be wider than the conv (class rows are padded to the 16-byte store width when nc is not a multiple
        # of it): the extra filters are zero and the extra channels are never read
        pk = self._packed(seq[2], None, x.device, x.dtype, False, pad_cout=int(out.shape[1]))
        hip_conv2d(t, pk, 1, 0, L.ACT_NONE, out=out)

    def _ncp(self, dtype) -> int:
        """Class channels as stored: nc rounded up to 16 bytes of `dtype` (any nc is accepted, as in the reference)."""
        e = 16 // (2 if dtype == torch.bfloat16 else 4)
        return (self.nc + e - 1) // e * e

    # ---- branch-level concurrency -------------------------------------------------------------------------------------
    # The box and class branches of a level are independent 3-conv chains and a level only depends on its own input
    # (head.py:116-126), so the executor may start a level the moment its feature map exists (`start_level`, called by
    # BaseModel._predict_once) on two side HIP streams forked from the producing stream; `forward` joins them before
    # the decode.  Under hipGraph capture the event fork/join becomes parallel graph branches.  The small 20x20/40x40
    # launches of the neck and the head are latency-bound (a few hundred workgroups on 256 CUs), so they overlap well.
    concurrent = True

    def _side_streams(self, device):
        pools = self.__dict__.setdefault("_streams", {})
        key = (R.current_tag(), str(device))
        st = pools.get(key)
        if st is None:
            st = [torch.cuda.Stream(device=device) for _ in range(2 * self.nl)]
            pools[key] = st
        return st

    def _pend(self):
        return self.__dict__.setdefault("_pending", {}).setdefault(R.current_tag(), {})

    def start_level(self, i: int, x: torch.Tensor, defer_ok: bool = True) -> None:
        if self._keys_only() and (R.current_opts() is None or not R.current_opts().keys_only):
            with R.use_opts(keys_only=1):
                return self._start_level(i, x, defer_ok)
        return self._start_level(i, x, defer_ok)

    def _start_level(self, i: int, x: torch.Tensor, defer_ok: bool = True) -> None:
        """Launch level i's two branches (asynchronously when `concurrent`); results land in the level's raw buffer and /
        or, with the fused decode, directly in the decoded output.  On one stream (linear graphs) the levels are left to `forward`,
        which runs them several per launch (`_levels_grouped`)."""
        pend = self._pend()
        x = R.to_nhwc(x, x.dtype)
        nb = 4 * self.reg_max
        n, _, h, w = x.shape
        plan = self._plan().get(R.current_tag())
        if plan is not None and (plan["n"] != n or plan["hw"][i] != (h, w) or plan["y"].device != x.device):
            plan = None  # stale plan (other input size): this level goes through the separate decode
        fused = plan is not None and plan["fused"] and x.dtype == torch.bfloat16
        ncp = self._ncp(x.dtype)
        buf = None
        if not fused or self.keep_raw:
            buf = R.alloc_nhwc(n, nb + ncp, h, w, x.dtype, x.device, key=(id(self), "raw", i))
        outs = (None, None) if buf is None else (buf[:, :nb], buf[:, nb:])
        raw = None if buf is None else buf[:, :self.no]

        def run(k):
            seq = (self.cv2[i], self.cv3[i])[k]
            if fused:
                self._tail(seq, x, outs[k], k + 1, i, plan)
            else:
                self._branch(seq, x, outs[k])

        linear = not self.concurrent or R.current_tag() != 0
        if defer_ok and linear and fused and self.group_levels and self.fuse_branch and not self.keep_raw:
            return  # `forward` picks it up (grouped with the other small levels)
        if fused and buf is None and self.fuse_branch and self._level_stream(i, x, plan):
            # the same line-buffer launch on every walk (grouped, level by level, concurrent): the output does not depend on the walk
            plan["decoded"].add(i)
            pend[i] = (None, [])
            return
        if fused:
            plan["decoded"].add(i)
        if linear:
            # inside a concurrently scheduled sub-batch (BaseModel.compile(micro_batches>1)) the branches stay on the
            # sub-batch's stream: the sub-batches already overlap each other, and nesting a second level of event
            # forks inside a forked capture stream crashed hipStreamEndCapture on ROCm 7.2 (segfault, not an error code)
            run(0)
            run(1)
            pend[i] = (raw, [])
            return
        main = torch.cuda.current_stream(x.device)
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []
        streams = self._side_streams(x.device)
        for k in range(2):
            side = streams[2 * i + k]
            side.wait_event(fork)
            with torch.cuda.stream(side):
                run(k)
                ev = torch.cuda.Event()
                ev.record(side)
            joins.append(ev)
        pend[i] = (raw, joins)

    def forward(self, x):
        if self._keys_only() and (R.current_opts() is None or not R.current_opts().keys_only):
            with R.use_opts(keys_only=1):
                return self._forward(x)
        return self._forward(x)

    def _forward(self, x):
        if self.training:
            raise L.UpaError("training-mode Detect is not on the HIP path yet (SURVEY §8f rank 2)")
        pend = self._pend()
        if not pend:  # called directly (not through BaseModel._predict_once): the level shapes come with the inputs
            self.begin(x[0].shape[0], [(t.shape[2], t.shape[3]) for t in x], x[0].dtype, x[0].device)
        rest = [i for i in range(self.nl) if i not in pend]
        plan = self._plan().get(R.current_tag())
        linear = not self.concurrent or R.current_tag() != 0
        grp = list(rest)
        if (len(grp) >= 2 and linear and self.group_levels and self.fuse_branch and not self.keep_raw and plan is not None and plan["fused"]
                and all(x[i].dtype == torch.bfloat16 and plan["n"] == x[i].shape[0] and plan["hw"][i] == tuple(x[i].shape[2:])
                        and plan["y"].device == x[i].device for i in grp)):
            xs_n = [R.to_nhwc(x[i], x[i].dtype) for i in grp]
            streamed = self._level_stream(grp[0], xs_n[0], plan)  # the first (largest) level as ONE line-buffer launch where it applies
            if streamed:
                plan["decoded"].add(grp[0])
                pend[grp[0]] = (None, [])
                grp, xs_n = grp[1:], xs_n[1:]
            if len(grp) >= 2 and self._levels_grouped(grp, xs_n, plan, allow_stack=not streamed):
                for i in grp:
                    plan["decoded"].add(i)
                    pend[i] = (None, [])
        for i in range(self.nl):
            if i not in pend:
                self.start_level(i, x[i], defer_ok=False)
        main = torch.cuda.current_stream(x[0].device)
        raw = []
        for i in range(self.nl):
            buf, joins = pend[i]
            for ev in joins:
                main.wait_event(ev)
            raw.append(buf)
        pend.clear()
        y = self._inference(raw)
        return y if self.export else (y, raw)

    def _inference(self, x: list) -> torch.Tensor:
        """Decode boxes and class probabilities of all levels into (B, 4+nc, A) float32 (head.py:151-169).  Levels whose
        branches already decoded in their last conv (`upa_detect_tail`) are skipped; x[i] may then be None."""
        nb = 4 * self.reg_max
        plan = self._plan().pop(R.current_tag(), None)
        done = plan["decoded"] if plan is not None else set()
        if plan is not None and (done or all(t is not None for t in x)):
            y, a_total = plan["y"], plan["a_total"]
        else:
            n = x[0].shape[0]
            a_total = sum(int(t.shape[2]) * int(t.shape[3]) for t in x)
            y = R.alloc_plain((n, 4 + self.nc, a_total), torch.float32, x[0].device, key=(id(self), "y"))
        a0 = 0
        for i, t in enumerate(x):
            if i in done:
                a0 += plan["hw"][i][0] * plan["hw"][i][1]
                continue
            vb, vc = R.view_of(t[:, :nb]), R.view_of(t[:, nb:])
            L.check(L.lib().upa_detect_decode(vb.ptr, vb.ld, vc.ptr, vc.ld, vb.n, vb.h, vb.w, self.reg_max, self.nc,
                                              float(self.stride[i]), y.data_ptr(), a_total, a0, vb.dtype,
                                              L.current_stream(t.device)), "detect_decode")
            a0 += vb.h * vb.w
        # NMS prefilter: valid only when every level's class branch wrote its keys in THIS forward
        hot = plan.get("hot") if plan is not None else None
        if hot is not None and plan["hot_levels"] == set(range(self.nl)) and y is plan["y"]:
            y._upa_hot = hot
        elif hasattr(y, "_upa_hot"):
            del y._upa_hot
        # keys-only mode (`scores_out = False`): rows 4.. of y were NOT written in this forward - mark the tensor so that every reader of
        # class scores (multi-label NMS, the validator) refuses it instead of reading stale memory
        if self._keys_only() and hasattr(y, "_upa_hot"):
            y._upa_keys_only = True
        elif hasattr(y, "_upa_keys_only"):
            del y._upa_keys_only
        return y

    def bias_init(self):
        """Initialize Detect() biases, requires stride availability (head.py:171-178)."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / s) ** 2)

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)
