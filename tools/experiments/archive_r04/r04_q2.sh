#!/bin/bash
# kernel-level split of one weight-gradient call (main kernel vs partial-sum reduce) for a few layer shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "64 64 3 1 80 32" "256 256 3 1 20 32" "128 128 3 1 80 32" "768 512 1 1 20 32" "96 64 1 1 160 32" "32 64 3 2 320 32"; do
  d=$R/gpurun_out/q2/$(echo $cfg | tr ' ' '_')
  UPA_HIP_LIB=$R/ultralytics_pro_amd/${LIBF:-libupa_hip.so} rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/tools/bench_wgrad.py $cfg > /dev/null 2>&1
  echo "== $cfg"
  python3 - $d <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "wgrad" in r["Name"]:
        print("  %-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
