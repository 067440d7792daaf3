#!/bin/bash
# persistent-kernel grid sizes vs throughput of the pipelined default run
b() { echo "$* : $(env "$@" python bench.py --opts env --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"; }
b A=1
b A=1
b UPA_PIPE_WGS=128
b UPA_PIPE_WGS=512
b UPA_STEMF_WGS=256
b UPA_STEMF_WGS=1024
b UPA_C1_WGS=128
b UPA_C1_WGS=64
b UPA_PIPE_WGS=128 UPA_STEMF_WGS=256 UPA_C1_WGS=128
b UPA_CONV_NO_WS=1
b UPA_CONV_NO_PIPE=1
b UPA_CONV_NO_1X1=1
