#!/bin/bash
# FETCH_SIZE calibration for this library's access shapes (tools/experiments/fetch_calib.hip): bytes requested vs FETCH_SIZE
# reported, per run length.  usage (GPU box): tools/pmc_fetch_calib.sh  -> gpurun_out/fetch_calib/summary.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/fetch_calib
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $root/tools/experiments/fetch_calib.hip -o $out/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
$out/fetch_calib > $out/plain.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc -- $out/fetch_calib > $out/profiled.txt 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(f"{out}/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "read_runs" in r["Kernel_Name"]:
            rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
with open(f"{out}/summary.txt", "w") as fo:
    print(open(f"{out}/plain.txt").read(), file=fo)
    print("kernel, FETCH_SIZE (KB), FETCH_SIZE bytes / bytes requested (1 GiB)", file=fo)
    for k, v in rows:
        print(f"{k}  {v:.0f} KB  ratio {v * 1024 / (1 << 30):.3f}", file=fo)
print(open(f"{out}/summary.txt").read())
PY
