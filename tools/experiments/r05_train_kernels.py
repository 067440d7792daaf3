"""Per-step kernel table of a rocprofv3 --kernel-trace --stats run of `bench.py --workload train` (steps told by the optimizer launches).
    python3 tools/experiments/r05_train_kernels.py <dir> [rows]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
sg = [r for r in rows if "sgd_nesterov" in r["Name"]][0]
steps = int(sg["Calls"]) / 3
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"{steps:.0f} steps, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches / step, {tot / steps / 1e6:.2f} ms of kernel time / step")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']) / steps:6.1f} x {float(r['AverageNs']) / 1e3:7.1f} us = {int(r['TotalDurationNs']) / steps / 1e3:7.1f} us / step")
