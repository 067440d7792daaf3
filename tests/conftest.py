import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"

# The software-pipelined 3x3 kernel (csrc/conv_pipe.hip) is only dispatched to layers with >= 1024 wave tiles in
# production; the parity tests run it on every eligible shape (read once by the library at first use).
import os  # noqa: E402

os.environ.setdefault("UPA_PIPE_MIN_TILES", "1")
os.environ.setdefault("UPA_PIPE_ALL", "1")
# the fused Bottleneck kernel (csrc/conv_pair.hip) is dispatched for C = 32 only in production (the 64-channel form is slower
# than two launches at 40x40); the parity tests run both widths
os.environ.setdefault("UPA_NO_PAIR", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
