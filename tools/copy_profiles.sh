#!/bin/bash
# Copy the summaries of gpurun_out/final_<round>/ (tools/round_profiles.sh) into profiles/ as <round>_*.  Adds files of the CURRENT
# round only; refuses to overwrite a file of another round, and never edits profiles/README.md (rows of earlier rounds are history).
R=$(cd "$(dirname "$0")/.." && pwd)
ROUND=${ROUND:-r05}
O=$R/gpurun_out/final_$ROUND
P=$R/profiles
cpf() { [ -f "$1" ] && cp "$1" "$P/${ROUND}_$2" && echo "  ${ROUND}_$2"; }
cpf $O/bench_default.json bench_default_bf16.json
cpf $O/bench_default_with_traffic.json bench_default_bf16_with_traffic.json
cpf $O/bench_serial.json bench_serial_bf16.json
cpf $O/bench_serial_profiled.json bench_serial_profiled_bf16.json
cpf $O/bench_f32.json bench_default_f32.json
cpf $O/bench_train.json bench_train_bf16.json
cpf $O/bench_val.json bench_val_bf16.json
cpf $O/bench_yolov3-rtdetr_serial.json bench_yolov3-rtdetr_serial.json
cpf $O/wgrad_layers_yolov8s.txt wgrad_layers_yolov8s.txt
cpf $O/bench_default_no_mode_dispatch.json bench_default_no_mode_dispatch.json
for m in yolov8s yolov3-tiny yolov5-BoT3 yolov5-BoT3_bs32 yolov3-rtdetr; do cpf $O/bench_$m.json bench_$m.json; done
for m in yolov8n yolov8s yolov3-tiny yolov3-rtdetr; do cpf $O/conv_layers_$m.txt conv_layers_$m.txt; done
cpf $O/pmc_hbm_summary.json pmc_hbm_summary.json
cpf $O/pmc_wgrad_summary.json pmc_wgrad_summary.json
cpf $O/pmc_step_summary.txt pmc_step_budget.txt
for k in default serial train val; do
  f=$(ls -t $(find $O/prof_$k -name "*kernel_stats.csv" 2>/dev/null) 2>/dev/null | head -1)   # the newest: gpurun merges a call's files into the directory, traces of earlier calls stay
  [ -n "$f" ] && cpf "$f" bench_${k}_kernel_stats.csv
done
