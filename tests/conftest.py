import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"

# Dispatch options of the test session (upa_opts, include/upa.h; the library itself reads no environment variable):
#  * the software-pipelined 3x3 kernel (csrc/conv_pipe.hip) is only dispatched to layers with >= 1024 wave tiles in production;
#    the parity tests run it on every eligible shape;
#  * the fused Bottleneck kernel (csrc/conv_pair.hip) is dispatched for C = 32 only in production (the 64-channel form is slower
#    than two launches at 40x40); the parity tests run both widths.
from ultralytics_pro_amd import _lib as _L  # noqa: E402
from ultralytics_pro_amd.engine import runtime as _R  # noqa: E402

TEST_OPTS = _L.Opts(pipe_min_tiles=1, pipe_all=1, pair=2, c2f64_max_px=-1)
_R.set_default_opts(TEST_OPTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
