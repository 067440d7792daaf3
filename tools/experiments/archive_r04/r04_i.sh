#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "keys_only or nms_prefilter or full_size_properties or headline_batch" 2>&1 | grep -v amdgpu.ids | tail -8 > $O/tests.log
cat $O/tests.log
bash tools/experiments/ab_opts.sh "--full-scores" "" > $O/ab_keys_only.txt 2>&1
cat $O/ab_keys_only.txt
