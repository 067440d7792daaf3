#!/bin/bash
# HBM traffic per launch of the weight-gradient kernels of the yolov8s bs=32 training step from rocprofv3 PMC counters, collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes (never combined with trace domains), both in KB; on
# gfx950 FETCH_SIZE counts 64 B per 128-B request and is doubled.  The passes run tools/experiments/wgrad_pmc_table.py (two training steps, then
# every layer's upa_conv2d_wgrad re-issued alone): a family's mean is over its launches of the step's layer set, like
# roofline.algorithmic_flops_per_launch of `bench.py --workload train`.  Writes gpurun_out/pmc_wgrad/summary.json
# (copy to profiles/<round>_pmc_wgrad_summary.json).   usage (GPU box): tools/pmc_wgrad.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_wgrad
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pm in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pm --output-format csv -d $out/$pm -- python3 $root/tools/experiments/wgrad_pmc_table.py > /dev/null 2>&1
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for pm in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{pm}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == pm and "wgrad" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][pm].append(float(r["Counter_Value"]))
kern = {}
for k, d in acc.items():
    if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
        continue
    fm, wm = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    kern[k] = {"FETCH_SIZE_KB_mean": fm, "FETCH_SIZE_n": len(d["FETCH_SIZE"]), "WRITE_SIZE_KB_mean": wm,
               "WRITE_SIZE_n": len(d["WRITE_SIZE"]), "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0}
summary = {"command": "tools/pmc_wgrad.sh = rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- "
                      "python3 tools/experiments/wgrad_pmc_table.py",
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md HBM); WRITE_SIZE exact; both in KB",
           "config": "yolov8s bs=32 bf16 train", "kernels": kern}
json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)
for k, v in sorted(kern.items()):
    print(f"{k[:90]:90s} n={v['FETCH_SIZE_n']:4d} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")
PY
