#!/bin/bash
# HBM traffic per kernel launch of `bench.py --serial` from rocprofv3 PMC counters, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in separate passes (never combined with trace domains), both in KB; on gfx950 FETCH_SIZE
# counts 64 B per 128-B request and is doubled.  Writes gpurun_out/pmc_hbm/summary.json (copied to profiles/<round>_pmc_hbm_summary.json by tools/copy_profiles.sh).  The pass runs one step at a time (counter
# collection serialises kernels anyway) but with the kernel choice of the headline run's in-flight copies (upa_opts c2f=4, conv_ws3=1, c2f_stream_rows=-1, detect_stream=2, conv_big=2:
# engine/pipeline.py), so that a family's per-launch average is over the launches the timed region replays.
# usage (GPU box): tools/pmc_hbm.sh [extra bench.py args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_hbm
cd /tmp && export TMPDIR=/tmp
for pm in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pm --output-format csv -d $out/$pm -- python3 $root/bench.py --serial --opts c2f=4,conv_ws3=1,c2f_stream_rows=-1,detect_stream=2,conv_big=2 --steps 3 --warmup 1 --input-batches 1 --no-cpu-baseline "$@" > /dev/null 2>&1
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for pm in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{pm}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == pm:
                acc[r["Kernel_Name"]][pm].append(float(r["Counter_Value"]))
kern = {}
for k, d in acc.items():
    if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
        continue
    fm, wm = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    kern[k] = {"FETCH_SIZE_KB_mean": fm, "FETCH_SIZE_n": len(d["FETCH_SIZE"]), "WRITE_SIZE_KB_mean": wm,
               "WRITE_SIZE_n": len(d["WRITE_SIZE"]), "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0}
summary = {"command": "tools/pmc_hbm.sh = rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- "
                      "python3 bench.py --serial --opts c2f=4,conv_ws3=1,c2f_stream_rows=-1,detect_stream=2,conv_big=2 --steps 3 --warmup 1 --input-batches 1 --no-cpu-baseline",
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md HBM); WRITE_SIZE exact; both in KB",
           "config": "yolov8n bs=32 bf16", "kernels": kern}
import os
rnd = os.environ.get("ROUND", "r04")
json.dump(summary, open(f"profiles/{rnd}_pmc_hbm_summary.json", "w"), indent=1)  # so that the bench runs after this one in the same call see it
json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)  # gpurun merges gpurun_out/ back, not profiles/
print("kernels:", len(kern))
PY
