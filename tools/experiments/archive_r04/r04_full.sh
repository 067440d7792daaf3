#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04full; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15 > $O/tests.log
cat $O/tests.log
