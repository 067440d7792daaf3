#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04d; mkdir -p $O
python tools/experiments/archive_r04/r04_diag.py 2>&1 | grep -v amdgpu.ids > $O/diag.txt
timeout 900 python -m pytest tests -m gpu -q -s -k "rounding_point_emulation" 2>&1 | grep -v amdgpu.ids > $O/emul.log
cat $O/diag.txt; grep -E "vs emulation|passed|failed" $O/emul.log | cut -c1-900
