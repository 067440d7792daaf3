#!/bin/bash
# first-run effect on a fresh box: usage r04_f2.sh "<args of the first run>"
cd "$(dirname "$0")/../.." || exit 1
j() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('serial_ms_per_step'))"; }
echo "first: $1"; python bench.py --no-cpu-baseline --no-kernel-profile --no-parity $1 2>/dev/null | j
echo "second: default"; python bench.py --no-cpu-baseline --no-kernel-profile --no-parity 2>/dev/null | j
echo "third: --steps 20 --warmup 5"; python bench.py --no-cpu-baseline --no-kernel-profile --no-parity --steps 20 --warmup 5 2>/dev/null | j
echo "fourth: --steps 20 --warmup 2000"; python bench.py --no-cpu-baseline --no-kernel-profile --no-parity --steps 20 --warmup 2000 2>/dev/null | j
