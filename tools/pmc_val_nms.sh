#!/bin/bash
# HBM traffic per launch of the validate call's NMS kernels from rocprofv3 PMC counters (FETCH_SIZE / WRITE_SIZE in separate passes, never
# combined with trace domains; both in KB; gfx950: FETCH_SIZE counts 64 B per 128-B request and is doubled - as tools/pmc_hbm.sh).  The
# program under the profiler is tools/experiments/r05_val_nms_counts.py (forward + one NMS call per batch).  Prints and writes
# gpurun_out/pmc_val_nms/summary.txt.     usage (GPU box): tools/pmc_val_nms.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_val_nms
cd /tmp && export TMPDIR=/tmp
for pm in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $pm --output-format csv -d $out/$pm -- python3 $root/tools/experiments/r05_val_nms_counts.py --batches 4 > /dev/null 2>&1
done
cd $root
python3 - "$out" <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for pm in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{pm}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == pm and ("nms" in r["Kernel_Name"] or "zero_words" in r["Kernel_Name"]):
                acc[r["Kernel_Name"].split("(")[-2 if r["Kernel_Name"].startswith("(") else 0][:40] if False else r["Kernel_Name"][:60]][pm].append(float(r["Counter_Value"]))
print("validate-call NMS kernels, yolov8n bs 32 (B = 32, A = 8400, nc = 80: 86.0 MB of f32 class scores per batch); MB per launch, mean (max)")
for k, d in sorted(acc.items()):
    f, w = d["FETCH_SIZE"], d["WRITE_SIZE"]
    if not f or not w:
        continue
    print(f"{k:62s} launches {len(f):3d}  read {2 * sum(f) / len(f) / 1024:8.2f} ({2 * max(f) / 1024:8.2f})  written {sum(w) / len(w) / 1024:8.2f} ({max(w) / 1024:8.2f})")
PY
