"""Tensor plumbing between the Python operator API and the C ABI.

Activations are ordinary torch tensors whose LOGICAL shape is the reference's (N, C, H, W) but whose memory is an
NHWC "view": strides (H*W*ld, 1, W*ld, ld) with ld >= C, i.e. possibly a channel slice of a wider concat buffer.
torch is used for allocation, streams and views only - no torch kernel touches activation data on the product path.
"""

from __future__ import annotations

import contextlib
import ctypes as C
from dataclasses import dataclass

import torch

from .. import _lib as L


# ---- dispatch options of the calls being made (upa_opts, include/upa.h) -------------------------------------------------
# The C library keeps no mode: every entry point that dispatches among kernel families takes a caller-owned `upa_opts*`.
# On the Python side the options in force are a stack PER THREAD (upa.h promises per-thread re-entrancy, and two models walked by two
# Python threads must not see each other's pushes or pop each other's entries): BaseModel._predict_once pushes its model's `opts`
# (two models in one process can differ), the parity tests and sweep tools wrap calls in `use_opts(L.Opts(...))`; below every
# thread's stack sits the process-wide `_DEFAULT_OPTS` (None = the library defaults; tests/conftest.py sets one so that every
# kernel family sees small shapes).
import threading

_TLS = threading.local()
_DEFAULT_OPTS = [None]


def _stack():
    st = getattr(_TLS, "opts", None)
    if st is None:
        st = _TLS.opts = []
    return st


def set_default_opts(opts):
    """Process-wide default of THIS PYTHON PACKAGE (not of the C library): the options used when no `use_opts` is active."""
    _DEFAULT_OPTS[0] = opts


def current_opts():
    st = _stack()
    return st[-1] if st else _DEFAULT_OPTS[0]


def opts_ptr():
    """`const upa_opts*` argument for a C call: pointer to the options in force, or None (NULL = defaults)."""
    o = current_opts()
    return None if o is None else C.pointer(o)


@contextlib.contextmanager
def use_opts(opts=None, **fields):
    """Run the enclosed calls of THIS thread with `opts` (an `_lib.Opts`), or with the options in force changed by `fields`."""
    if fields:
        base = opts if opts is not None else current_opts()
        opts = (base.replace(**fields) if base is not None else L.Opts(**fields))
    st = _stack()
    st.append(opts)
    try:
        yield opts
    finally:
        st.pop()


@dataclass
class View:
    ptr: int
    n: int
    h: int
    w: int
    c: int
    ld: int
    dtype: int  # UPA_F32 / UPA_BF16


def _pixel_stride(t: torch.Tensor) -> int:
    n, c, h, w = t.shape
    s = t.stride()
    if w > 1:
        return s[3]
    if h > 1:
        return s[2]
    if n > 1:
        return s[0]
    return c


def is_nhwc_view(t: torch.Tensor) -> bool:
    """True iff t is logically (N, C, H, W) with strides (H*W*ld, 1, W*ld, ld), ld >= C (size-1 dims are free)."""
    if t.dim() != 4:
        return False
    n, c, h, w = t.shape
    s = t.stride()
    ld = _pixel_stride(t)
    return (ld >= c and (c == 1 or s[1] == 1) and (w == 1 or s[3] == ld) and (h == 1 or s[2] == w * ld)
            and (n == 1 or s[0] == h * w * ld))


def view_of(t: torch.Tensor) -> View:
    """Describe an NHWC-strided logical-NCHW tensor for the C ABI (raises if the layout is anything else)."""
    L.require_gpu(t, "view_of")
    if getattr(t, "_upa_pad_br", 0):
        raise L.UpaError("a ZeroPad2d-tagged tensor can only feed MaxPool2d on the HIP path")
    if not is_nhwc_view(t):
        raise L.UpaError(f"tensor with shape {tuple(t.shape)} strides {t.stride()} is not an NHWC view")
    n, c, h, w = t.shape
    return View(t.data_ptr(), n, h, w, c, _pixel_stride(t), L.dtype_code(t.dtype))


_TRACE_POOL = bool(__import__("os").environ.get("UPA_POOL_TRACE"))


class BufferPool:
    """Static activation buffers keyed by call site: the second and later forwards of a model allocate nothing, which
    is what makes the layer loop capturable into a hipGraph (no allocator calls inside the captured region)."""

    def __init__(self):
        self.buffers: dict = {}

    def get(self, key, shape, dtype, device) -> torch.Tensor:
        k = (key, tuple(shape), dtype, str(device))
        t = self.buffers.get(k)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=device)
            self.buffers[k] = t
            if _TRACE_POOL:
                import sys
                print(f"[pool] {t.data_ptr():#x} +{t.numel() * t.element_size():#x} end {t.data_ptr() + t.numel() * t.element_size():#x} {key}",
                      file=sys.stderr, flush=True)
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.buffers.values())


_POOL: list = [None]
_TAG: list = [0]  # micro-batch tag: keeps the static buffers / side streams of concurrently scheduled sub-batches apart


@contextlib.contextmanager
def pool_tag(tag):
    prev = _TAG[0]
    _TAG[0] = tag
    try:
        yield
    finally:
        _TAG[0] = prev


def current_tag():
    return _TAG[0]


@contextlib.contextmanager
def static_buffers(pool: BufferPool):
    prev = _POOL[0]
    _POOL[0] = pool
    try:
        yield pool
    finally:
        _POOL[0] = prev


def _unkeyed(what):
    """An allocation without a key while static buffers are active would be freed after graph capture and leave the
    captured launches with a dangling pointer: refuse it."""
    raise L.UpaError(f"unkeyed allocation {what} inside static_buffers(): every buffer of a captured step needs a key")


def alloc_nhwc(n: int, c: int, h: int, w: int, dtype: torch.dtype, device, key=None) -> torch.Tensor:
    """New NHWC buffer returned as a logical (N, C, H, W) tensor."""
    pool = _POOL[0]
    if pool is not None and key is not None:
        buf = pool.get((_TAG[0], key), (n, h, w, c), dtype, device)
    else:
        if pool is not None:
            _unkeyed(("nhwc", n, c, h, w))
        buf = torch.empty((n, h, w, c), dtype=dtype, device=device)
    return buf.permute(0, 3, 1, 2)


def alloc_plain(shape, dtype, device, key=None) -> torch.Tensor:
    pool = _POOL[0]
    if pool is not None and key is not None:
        return pool.get((_TAG[0], key), tuple(shape), dtype, device)
    if pool is not None:
        _unkeyed(("plain",) + tuple(shape))
    return torch.empty(tuple(shape), dtype=dtype, device=device)


def channel_slice(t: torch.Tensor, c0: int, c1: int) -> torch.Tensor:
    """Channel sub-view [c0, c1) of an NHWC-strided tensor (pure view arithmetic)."""
    return t[:, c0:c1]


def to_nhwc(x: torch.Tensor, dtype: torch.dtype, key=None) -> torch.Tensor:
    """Bring an arbitrary 4-D tensor into an NHWC view of `dtype` (HIP transpose kernel when a copy is needed)."""
    L.require_gpu(x, "to_nhwc")
    if is_nhwc_view(x) and x.dtype == dtype and (x.shape[1] * x.element_size()) % 16 == 0 and x.data_ptr() % 16 == 0:
        return x
    if x.dtype != torch.float32 or not x.is_contiguous():
        if is_nhwc_view(x):  # NHWC of another dtype: go through NCHW f32
            x = to_nchw_f32(x)
        else:
            raise L.UpaError("to_nhwc expects a contiguous float32 NCHW tensor or an NHWC view")
    n, c, h, w = x.shape
    y = alloc_nhwc(n, c, h, w, dtype, x.device, key)
    v = view_of(y)
    L.check(L.lib().upa_nchw_to_nhwc(x.data_ptr(), n, c, h, w, v.ptr, v.ld, v.dtype, L.current_stream(x.device)),
            "nchw_to_nhwc")
    return y


def to_nchw_f32(x: torch.Tensor, key=None) -> torch.Tensor:
    """NHWC view -> contiguous float32 NCHW (the reference's tensor layout), via the HIP transpose kernel."""
    v = view_of(x)
    y = alloc_plain((v.n, v.c, v.h, v.w), torch.float32, x.device, key)
    L.check(L.lib().upa_nhwc_to_nchw(v.ptr, v.n, v.h, v.w, v.c, v.ld, y.data_ptr(), v.dtype, L.current_stream(x.device)),
            "nhwc_to_nchw")
    return y


class HipGraph:
    """Capture a sequence of upa_* launches issued on the current stream and replay it (upa_graph_* in upa.h)."""

    def __init__(self):
        self.exec = C.c_void_p()
        self.stream = None

    def capture(self, fn, device=None):
        self.stream = torch.cuda.Stream(device=device)
        self.stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(self.stream):
            s = self.stream.cuda_stream
            L.check(L.lib().upa_graph_begin(s), "graph_begin")
            try:
                out = fn()
            finally:
                rc = L.lib().upa_graph_end(s, C.byref(self.exec))
            L.check(rc, "graph_end")
        torch.cuda.current_stream(device).wait_stream(self.stream)
        return out

    def replay(self, device=None):
        """Launch the instantiated graph in stream order with the caller's current stream.

        Never on the null stream: replaying a graph with parallel branches there and graphs on other streams afterwards
        ended in GPU memory-access faults on ROCm 7.2 (tools/experiments/fault_bisect.sh: 1-2 of 6 processes survive;
        6 of 6 with every launch on a created stream) - from the null stream the launch goes to the capture stream,
        ordered before and after by stream waits."""
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream == 0:
            self.stream.wait_stream(cur)
            L.check(L.lib().upa_graph_launch(self.exec, self.stream.cuda_stream), "graph_launch")
            cur.wait_stream(self.stream)
        else:
            L.check(L.lib().upa_graph_launch(self.exec, cur.cuda_stream), "graph_launch")

    def __del__(self):
        try:
            if self.exec:
                L.lib().upa_graph_destroy(self.exec)
        except Exception:
            pass
