#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04w; mkdir -p $O
bash tools/experiments/ab_opts.sh "" "--opts stemf_wgs=256" "--opts stemf_wgs=384" 2>&1 | tee $O/ab_stemf_wgs.txt
