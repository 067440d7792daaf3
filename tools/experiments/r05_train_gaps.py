"""Where the training step's time goes on the GPU: from a rocprofv3 --kernel-trace CSV of `bench.py --workload train`, per HIP stream (the trace's
Stream_Id; a stream hops between hardware queues), the busy time, and the idle gaps between consecutive kernels of the busiest stream (the main one) over the last steps.
    python3 tools/experiments/r05_train_gaps.py <dir with *_kernel_trace.csv> [steps]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# step boundaries: the sgd kernel runs 3 times at the end of every step
sgd = [i for i, r in enumerate(rows) if "sgd_nesterov" in r["Kernel_Name"]]
ends = sgd[2::3]
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
sel = rows[lo:hi]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
print(f"{steps} steps, {len(sel) / steps:.0f} launches per step, {(t1 - t0) / steps / 1e6:.3f} ms per step on the GPU timeline")
byq = defaultdict(list)
for r in sel:
    byq[r.get("Stream_Id", r["Queue_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in v)
    print(f"stream {q}: {len(v) / steps:6.1f} launches / step, busy {busy / steps / 1e6:6.3f} ms / step")
q, v = max(byq.items(), key=lambda kv: len(kv[1]))
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(v, v[1:]):
    gaps.append((max(0, s1 - e0), n0, n1))
tot = sum(g for g, _, _ in gaps)
print(f"main stream {q}: idle between kernels {tot / steps / 1e6:.3f} ms / step; gaps > 20 us: {sum(1 for g, _, _ in gaps if g > 20000) / steps:.1f} / step "
      f"({sum(g for g, _, _ in gaps if g > 20000) / steps / 1e6:.3f} ms), 5-20 us: {sum(1 for g, _, _ in gaps if 5000 < g <= 20000) / steps:.1f} / step "
      f"({sum(g for g, _, _ in gaps if 5000 < g <= 20000) / steps / 1e6:.3f} ms), < 5 us: {sum(g for g, _, _ in gaps if g <= 5000) / steps / 1e6:.3f} ms")
import collections
big = collections.Counter()
for g, n0, n1 in gaps:
    if g > 20000:
        big[(n0[:50], n1[:50])] += g
for (a, b), g in big.most_common(8):
    print(f"   {g / steps / 1e3:8.1f} us / step   after {a}  before {b}")
if len(sys.argv) > 3:  # details of the largest gaps: what the other queues ran meanwhile
    allk = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r["Queue_Id"]), r["Kernel_Name"]) for r in sel]
    big = sorted(((s1 - e0, e0, s1, n0, n1) for (s0, e0, n0), (s1, e1, n1) in zip(v, v[1:])), reverse=True)[:int(sys.argv[3])]
    for g, a, b, n0, n1 in big:
        others = [(s, e, qq, n) for s, e, qq, n in allk if qq != q and e > a and s < b]
        print(f"gap {g / 1e3:7.1f} us  after {n0[:60]}  before {n1[:60]}")
        for s, e, qq, n in others[:6]:
            print(f"      queue {qq}: {n[:70]} [{(s - a) / 1e3:7.1f} .. {(e - a) / 1e3:7.1f}] us rel. to the gap start")
