"""Phase times of the workgroups of ONE launch of a kernel built with -DUPA_STAMP (csrc/common.h: UPA_STAMP_AT): wave 0 of each
workgroup records s_memtime at its phase boundaries.  Needs UPA_HIP_LIB=<library built with -DUPA_STAMP>."""
import ctypes as C

import numpy as np
import torch

from ultralytics_pro_amd import _lib as L


def report(launch, tag, names, ticks_per_us=100.0):
    lib = L.lib()
    rd, clr = getattr(lib, f"upa_debug_stamps_{tag}"), getattr(lib, f"upa_debug_stamps_clear_{tag}")
    rd.argtypes = [C.c_void_p, C.c_int]
    launch()
    torch.cuda.synchronize()
    assert clr() == 0
    launch()
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 16, dtype=np.uint64)
    assert rd(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(4096, 16).astype(np.int64)
    nz = st[:, 0] > 0
    if nz.sum() == 0:
        print(f"  [{tag}] no stamps recorded (kernel not launched?)")
        return st
    first = placement(st[: int(nz.sum())])
    some = sorted(first.items())[:6]
    print("  workgroups per CU (first CUs):", "; ".join(f"{c}: {v}" for c, v in some))
    st = st[nz]
    last = len(names)
    t0 = st[:, 0].min()
    hw = st[:, 15]
    cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xF)  # xcc, se, cu
    span = 1
    print(f"  [{tag}] workgroups {len(st)}  distinct CUs {len(set(cu.tolist()))}")
    for k, nm in enumerate(names):
        d = st[:, k + 1] - st[:, k]
        print(f"  {nm:18s} median {int(np.median(d)):7d}  p10 {int(np.percentile(d, 10)):7d}  p90 {int(np.percentile(d, 90)):7d}")
    life = st[:, last] - st[:, 0]
    print(f"  workgroup life     median {int(np.median(life)):7d}  p10 {int(np.percentile(life, 10)):7d}  p90 {int(np.percentile(life, 90)):7d}"
          )
    # s_memtime is per XCD (the counters of different XCDs are millions of ticks apart): cluster the start stamps by value and
    # compare starts / ends within a cluster only
    order = np.argsort(st[:, 0])
    srt = st[order, 0]
    cl = np.concatenate([[0], np.cumsum(np.diff(srt) > 1_000_000)])
    start = np.zeros(len(st), dtype=np.int64)
    spans = []
    for c in range(cl.max() + 1):
        idx = order[cl == c]
        start[idx] = st[idx, 0] - st[idx, 0].min()
        spans.append(int(st[idx, last].max() - st[idx, 0].min()))
    print(f"  XCD clusters {cl.max() + 1}: first start -> last end per XCD, ticks: median {int(np.median(spans))} max {max(spans)}")
    print("  start times (ticks): histogram", np.histogram(start, bins=8)[0].tolist(), "max", int(start.max()))
    return st


def placement(st_all):
    """blockIdx -> (xcc, se, cu) placement table of the stamped launch: which workgroups shared a CU."""
    hw = st_all[:, 15]
    cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xF)
    by = {}
    for b, c in enumerate(cu.tolist()):
        by.setdefault(c, []).append(b)
    return by
