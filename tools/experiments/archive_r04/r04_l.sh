#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04l; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -5 > $O/tests.log
cat $O/tests.log
bash tools/experiments/ab_opts.sh "--opts no_xcd=1" "--opts no_xcd=0" > $O/ab_xcd.txt 2>&1
cat $O/ab_xcd.txt
