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


def test_product_builds_on_cpu_but_refuses_to_run_there():
    from ultralytics_pro_amd._lib import UpaError
    from ultralytics_pro_amd.nn.tasks import DetectionModel
    m = DetectionModel("yolov8n.yaml")
    assert sum(p.numel() for p in m.parameters()) == 3157200
    with pytest.raises(UpaError):
        m(torch.zeros(1, 3, 64, 64))
