"""ORACLE tooling: import the read-only reference (/root/reference) with permissive stubs for the third-party
packages this image lacks (cv2, torchvision, timm, ...).  Used ONLY by oracle/gen_golden.py in the build container;
never imported on the GPU box (the reference does not exist there) and never by the product.
SURVEY.md Appendix A is the specification this file implements.
"""

from __future__ import annotations

import importlib.abc
import importlib.machinery
import importlib.metadata
import sys
import types

REF_ROOT = "/root/reference"
STUB_TOPLEVEL = {
    "cv2", "torchvision", "timm", "thop", "fvcore", "pywt", "antialiased_cnns", "ttach", "ipdb", "basicsr",
    "pytorch_wavelets", "fairscale", "mmcv", "mmengine", "natten", "mamba_ssm", "selective_scan_cuda",
    "selective_scan_cuda_core", "selective_scan_cuda_oflex", "depthwise_conv2d_implicit_gemm", "triton", "DCNv3",
    "DCNv4", "polars", "seaborn", "cpuinfo", "kornia", "huggingface_hub", "matplotlib", "PIL", "pandas", "scipy",
    "lap", "shapely", "albumentations", "pycocotools", "efficientnet_pytorch", "calflops", "ptflops",
}


def _make_any():
    import torch

    class _Meta(type(torch.nn.Module)):
        def __getattr__(cls, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return cls

    class _Any(torch.nn.Module, metaclass=_Meta):
        """Usable as a base class, a decorator, a constant and an instance."""

        def __new__(cls, *a, **k):
            if cls is _Any and len(a) == 1 and not k and (isinstance(a[0], type) or callable(a[0])) \
                    and not isinstance(a[0], torch.nn.Module):
                return a[0]  # pass-through decorator
            return super().__new__(cls)

        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x=None, *a, **k):
            return x

        def __call__(self, *a, **k):
            if len(a) == 1 and not k and (isinstance(a[0], type) or isinstance(a[0], types.FunctionType)):
                return a[0]
            return super().__call__(*a, **k)

    return _Any


class _StubModule(types.ModuleType):
    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []
        self.__version__ = "0.20.0"
        self.__file__ = f"<stub {name}>"

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.isupper() and self.__name__.split(".")[0] == "cv2":  # cv2.IMREAD_COLOR-style enums
            return 0
        return _StubFinder.any_cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    any_cls = None

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_TOPLEVEL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def import_reference():
    """Returns the imported `ultralytics.nn.tasks` module of the reference."""
    sys.dont_write_bytecode = True
    import torch  # noqa: F401  first, keeps the default thread count (ultralytics/__init__.py:10-11)

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _StubFinder.any_cls = _make_any()
    real = {"scipy", "pandas", "PIL", "matplotlib", "huggingface_hub"}
    for name in list(STUB_TOPLEVEL):
        if name in real:
            try:
                __import__(name)
                STUB_TOPLEVEL.discard(name)
            except Exception:
                pass
    sys.meta_path.append(_StubFinder())
    _orig_version = importlib.metadata.version

    def _version(name):
        try:
            return _orig_version(name)
        except importlib.metadata.PackageNotFoundError:
            return "0.20.0"

    importlib.metadata.version = _version
    import ultralytics.nn.tasks as tasks  # noqa

    sys.modules.pop("torchvision", None)  # NMS must take the TorchNMS.nms branch (utils/nms.py:151-156)
    for k in [k for k in sys.modules if k.startswith("torchvision.")]:
        sys.modules.pop(k, None)
    return tasks
