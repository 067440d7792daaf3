"""The legs of bench.py that are not the timed region: kernel profile / roofline (profile.py), GPU-vs-oracle parity of the bench line
(parity.py), the CPU baseline (cpu_baseline.py: the ONLY one that imports oracle/, and only inside the child process bench.py starts for it).
bench.py imports them; the driver contract (`python bench.py --gpus N --steps K --warmup W`, one JSON line) lives there."""
