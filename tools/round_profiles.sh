#!/bin/bash
# Regenerates the numbers profiles/ holds (run on the GPU box; outputs under gpurun_out/final_<round>/, copied into profiles/ as
# <round>_* by tools/copy_profiles.sh).  ROUND=r04 by default.  This script and the copy step only ever ADD files named after the
# current round: rows and files of earlier rounds in profiles/ (and in profiles/README.md) are history and are never rewritten.
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r05}
O=$R/gpurun_out/final_$ROUND
rm -rf $O/prof_default $O/prof_serial $O/prof_train $O/prof_val   # stale traces of earlier calls would shadow this one's stats
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --serial --no-cpu-baseline > $O/bench_serial.json 2>/dev/null
timeout 600 python bench.py --dtype f32 --no-cpu-baseline --steps 200 > $O/bench_f32.json 2>/dev/null
for m in yolov8s yolov3-tiny; do timeout 600 python bench.py --model $m --no-cpu-baseline --steps 300 > $O/bench_$m.json 2>/dev/null; done
timeout 600 python bench.py --model yolov5-BoT3 --batch 16 --no-cpu-baseline --steps 300 > $O/bench_yolov5-BoT3.json 2>/dev/null   # BASELINE config 4: bs 16
timeout 600 python bench.py --model yolov5-BoT3 --batch 32 --no-cpu-baseline --steps 300 > $O/bench_yolov5-BoT3_bs32.json 2>/dev/null   # (rounds 2-4 quoted bs 32)
timeout 900 python bench.py --model yolov3-rtdetr --batch 16 > $O/bench_yolov3-rtdetr.json 2>/dev/null   # with the CPU leg: its parity object needs the oracle's output
timeout 600 python bench.py --workload train > $O/bench_train.json 2> $O/bench_train.err
timeout 600 python bench.py --model yolov3-rtdetr --batch 16 --serial --no-cpu-baseline --no-kernel-profile > $O/bench_yolov3-rtdetr_serial.json 2>/dev/null
timeout 300 python tools/experiments/wgrad_pmc_table.py > $O/wgrad_layers_yolov8s.txt 2>/dev/null   # per-layer weight-gradient table (both floors)
timeout 600 python bench.py --workload val --steps 300 > $O/bench_val.json 2> $O/bench_val.err
timeout 600 python bench.py --no-cpu-baseline --no-kernel-profile --no-mode-dispatch > $O/bench_default_no_mode_dispatch.json 2>/dev/null   # A/B: the one-step-at-a-time kernels (c2f64, conv_ws3) kept with four steps in flight
timeout 900 bash tools/pmc_hbm.sh --no-kernel-profile > /dev/null 2>&1
cp $R/gpurun_out/pmc_hbm/summary.json $O/pmc_hbm_summary.json
timeout 600 bash tools/pmc_wgrad.sh > /dev/null 2>&1
cp $R/gpurun_out/pmc_wgrad/summary.json $O/pmc_wgrad_summary.json
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default_with_traffic.json 2>/dev/null   # roofline.traffic from the PMC summary written just above
timeout 1200 bash tools/pmc_step.sh gpurun_out/pmc_step > /dev/null 2>&1
cp $R/gpurun_out/pmc_step/summary.txt $O/pmc_step_summary.txt
for m in yolov8n yolov8s yolov3-tiny; do timeout 300 python tools/bench_conv.py --model $m > $O/conv_layers_$m.txt 2>/dev/null; done
timeout 300 python tools/bench_conv.py --model yolov3-rtdetr --batch 16 > $O/conv_layers_yolov3-rtdetr.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --no-kernel-profile --steps 200 --warmup 10 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py --serial --no-cpu-baseline --steps 60 --warmup 5 > $O/bench_serial_profiled.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 $R/bench.py --workload train --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_val -- python3 $R/bench.py --workload val --steps 100 > /dev/null 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete   # large; the stats CSVs are what profiles/ keeps
echo done
