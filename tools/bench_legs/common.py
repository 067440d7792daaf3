"""Constants shared by bench.py and its legs."""
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
BENCH_PY = ROOT / "bench.py"

GFLOP_PER_IMG = 8.744  # yolov8n @640, 2*MAC over all 64 Conv2d
# the other configs of BASELINE.json (same convention); the headline metric is always yolov8n
GFLOP_OTHER = {"yolov3-tiny": 19.002, "yolov8s": 28.603, "yolov5-BoT3": 7.882 + 0.041, "yolov3-rtdetr": 256.56 + 11.49}
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3  # MFMA f32
PEAK_HBM_GBS = 8000.0
