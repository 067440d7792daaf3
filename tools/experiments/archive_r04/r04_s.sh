#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04s; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "stem or conv_mm" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/tests.log
cat $O/tests.log
python tools/experiments/archive_r04/r04_s.py 2>/dev/null | tee $O/stem_ab.txt
bash tools/experiments/ab_opts.sh "" "--opts no_xcd=1" 2>&1 | tee $O/ab.txt
