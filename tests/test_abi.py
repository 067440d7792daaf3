"""CPU (-m "not gpu"): the C-ABI library builds, loads and exports every symbol include/upa.h declares; the ctypes
prototype table covers the header; the product path refuses to run without a GPU (no CPU fallback)."""

import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def _header_functions():
    text = (ROOT / "include" / "upa.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(upa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from ultralytics_pro_amd import _lib
    if not _lib.LIB_PATH.is_file():
        import __graft_entry__ as g
        g.build()
    handle = ctypes.CDLL(str(_lib.LIB_PATH))
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/upa.h but not exported"
    assert set(names) == set(_lib.PROTOTYPES), "ctypes prototype table and include/upa.h disagree"
    assert _lib.lib().upa_version() >= 1


def test_host_side_weight_packing_layout():
    """upa_pack_conv_weight is host code: check the documented fragment order without a GPU."""
    from ultralytics_pro_amd import _lib as L
    cout, cin, k = 20, 24, 3
    w = torch.arange(cout * cin * k * k, dtype=torch.float32).reshape(cout, cin, k, k)
    for code, E, esz in ((L.UPA_F32, 4, 4), (L.UPA_BF16, 8, 2)):
        nbytes = L.lib().upa_conv_packed_weight_bytes(cout, cin, k, code)
        ktch = 4 * E
        ktt, ntn = -(-cin // ktch), -(-cout // 16)
        assert nbytes == k * k * ktt * ntn * 1024
        host = torch.zeros(nbytes, dtype=torch.uint8)
        L.check(L.lib().upa_pack_conv_weight(w.data_ptr(), cout, cin, k, code, host.data_ptr()))
        vals = host.view(torch.float32) if code == L.UPA_F32 else host.view(torch.bfloat16).float()
        vals = vals.reshape(k * k, ktt, ntn, 4, 16, E)
        wref = w.to(torch.bfloat16).float() if code == L.UPA_BF16 else w
        for (tap, kt, nt, g, r, j) in [(0, 0, 0, 0, 0, 0), (4, 0, 1, 2, 3, 1), (8, ktt - 1, 0, 1, 15, E - 1)]:
            co, ci = nt * 16 + r, kt * ktch + g * E + j
            exp = float(wref[co, ci, tap // k, tap % k]) if (co < cout and ci < cin) else 0.0
            assert float(vals[tap, kt, nt, g, r, j]) == exp


def test_dispatch_options_travel_with_the_call_and_the_library_reads_no_environment(monkeypatch):
    """Round-2 review item 4: include/upa.h promises no global state.  The dispatch query (host logic only, no launch) must follow
    the caller's `upa_opts`, two different option sets must not influence each other, and UPA_* environment variables must be
    ignored by the library (they were process-global switches in rounds 1-2)."""
    import ctypes as C
    import subprocess
    from ultralytics_pro_amd import _lib as L
    lib = L.lib()
    assert lib.upa_opts_size() == C.sizeof(L.Opts)
    q = (32, 40, 40, 128, 64, 3, 1, 1, L.UPA_BF16)  # yolov8n Detect cv2[1][0]: 128 -> 64 3x3 at 40x40, bs 32
    big = lambda v: (v >> 23) & 1  # noqa: E731
    default = lib.upa_conv_variant(*q, None)
    assert big(default)
    never, always = L.Opts(conv_big=1), L.Opts(conv_big=2)
    assert not big(lib.upa_conv_variant(*q, C.pointer(never)))
    assert lib.upa_conv_variant(*q, None) == default                      # the previous call left no mode behind
    small = (1, 16, 16, 128, 64, 3, 1, 1, L.UPA_BF16)                     # too few pixels for the size rule ...
    assert not big(lib.upa_conv_variant(*small, None)) and big(lib.upa_conv_variant(*small, C.pointer(always)))  # ... forced
    v128 = lib.upa_conv_variant(*q, C.pointer(L.Opts(conv_big_bm=128)))
    v256 = lib.upa_conv_variant(*q, C.pointer(L.Opts(conv_big_bm=256)))
    assert (v128 & 15, v256 & 15) == (1, 2)                                # pixels per workgroup / 128
    short = L.Opts(conv_big=1)
    short.size = 8                                                         # an older caller whose struct ends after conv_big
    assert not big(lib.upa_conv_variant(*q, C.pointer(short)))
    monkeypatch.setenv("UPA_CONV_BIG", "0")                                # the round-2 switch for "never": must be ignored now
    monkeypatch.setenv("UPA_CONV_BIG_BM", "128")
    assert lib.upa_conv_variant(*q, None) == default
    src = (ROOT / "ultralytics_pro_amd" / "csrc")
    hits = subprocess.run(["grep", "-ln", "getenv", *[str(f) for f in sorted(src.glob("*.hip")) + sorted(src.glob("*.h"))]],
                          capture_output=True, text=True).stdout.split()
    assert hits == [], f"getenv in the library sources: {hits}"
    ws3 = lambda v: (v >> 24) & 1  # noqa: E731
    q64 = (32, 40, 40, 64, 64, 3, 1, 1, L.UPA_BF16)                        # 64 -> 64 3x3: the persistent kernel by default ...
    assert ws3(lib.upa_conv_variant(*q64, None)) and big(lib.upa_conv_variant(*q64, C.pointer(L.Opts(conv_ws3=1))))  # ... conv_big on request
    assert L.Opts.from_env({"UPA_NO_PAIR": "0", "UPA_C1_MT": "4", "UPA_CONV_FORCE": "4,1,2,4"}).pair == 2  # tool-side mapping only


def test_conv_p8_size_rule_is_host_logic():
    """The round-4 dispatch rule of the two-group phased kernel (csrc/conv_p8.hip: one-round layers of 256-pixel x 128-channel tiles with
    Cin >= 256) is pure host logic behind `upa_conv_variant`: the layers it was measured on pick it, their neighbours stay on conv_big,
    `upa_opts.conv_p8` = 1 / 2 switch it off / force it, and conv_mm is never chosen by default."""
    import ctypes as C
    from ultralytics_pro_amd import _lib as L
    lib = L.lib()
    p8 = lambda v: (v >> 26) & 1   # noqa: E731
    big = lambda v: (v >> 23) & 1  # noqa: E731
    mm = lambda v: (v >> 25) & 1   # noqa: E731
    one_round = [(16, 40, 40, 512, 256), (16, 20, 20, 512, 1024), (16, 40, 40, 768, 256), (32, 20, 20, 256, 512), (32, 40, 40, 256, 128)]
    others = [(16, 40, 40, 256, 512), (16, 80, 80, 256, 128), (16, 80, 80, 128, 256), (16, 20, 20, 1024, 512), (32, 40, 40, 128, 128)]
    for n, h, w, c1, c2 in one_round:
        v = lib.upa_conv_variant(n, h, w, c1, c2, 3, 1, 1, L.UPA_BF16, None)
        assert p8(v) and not mm(v), (n, h, w, c1, c2, hex(v))
        assert big(lib.upa_conv_variant(n, h, w, c1, c2, 3, 1, 1, L.UPA_BF16, C.pointer(L.Opts(conv_p8=1))))
    for n, h, w, c1, c2 in others:
        v = lib.upa_conv_variant(n, h, w, c1, c2, 3, 1, 1, L.UPA_BF16, None)
        assert big(v) and not p8(v) and not mm(v), (n, h, w, c1, c2, hex(v))
        assert p8(lib.upa_conv_variant(n, h, w, c1, c2, 3, 1, 1, L.UPA_BF16, C.pointer(L.Opts(conv_p8=2))))
    assert not p8(lib.upa_conv_variant(16, 40, 40, 512, 256, 3, 2, 1, L.UPA_BF16, C.pointer(L.Opts(conv_p8=2))))  # stride 2: never
    assert mm(lib.upa_conv_variant(16, 40, 40, 512, 256, 3, 1, 1, L.UPA_BF16, C.pointer(L.Opts(conv_mm=2, conv_p8=1))))


def test_product_builds_on_cpu_but_refuses_to_run_there():
    from ultralytics_pro_amd._lib import UpaError
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    m = DetectionModel("yolov8n.yaml")
    assert sum(p.numel() for p in m.parameters()) == 3157200
    with pytest.raises(UpaError):
        m(torch.zeros(1, 3, 64, 64))
