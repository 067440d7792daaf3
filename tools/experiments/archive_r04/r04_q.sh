#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s -k "layer_by_layer" 2>&1 | grep -v amdgpu.ids | grep -E "layer |passed|failed|Error" > $O/tests.log
cat $O/tests.log
