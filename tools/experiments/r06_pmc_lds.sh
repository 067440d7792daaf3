#!/bin/bash
# LDS pipe occupancy per kernel under the throughput runner's dispatch (one step at a time, so that a launch has the chip to itself): SQ_LDS_IDX_ACTIVE
# (LDS-array cycles, summed over the CUs), SQ_LDS_BANK_CONFLICT (the extra cycles among them) and GRBM_GUI_ACTIVE (the launch in cycles) per launch;
# "busy" = LDS_IDX_ACTIVE / 256 CUs / GRBM_GUI_ACTIVE.  usage (GPU box): tools/experiments/r06_pmc_lds.sh [bench opts]
out=gpurun_out/pmc_lds
root=${GRAFT_REPO_ROOT:-$(pwd)}
opts=${1:-c2f=4,conv_ws3=1,c2f_stream_rows=-1,detect_stream=2}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/$out
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $root/$out/p1 -- python3 $root/bench.py --serial --opts $opts --steps 2 --warmup 1 --input-batches 1 --no-cpu-baseline --no-kernel-profile > /dev/null 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": cnt[k] += 1
print(f"{'kernel':62s} calls   LDS_IDX_ACTIVE  BANK_CONFLICT  conflict%  INSTS_LDS   launch cycles  LDS busy%   INSTS_VALU")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_LDS_IDX_ACTIVE", 0))[:22]:
    c = max(cnt[k], 1); a = agg[k]
    act, conf, gui = a.get("SQ_LDS_IDX_ACTIVE", 0) / c, a.get("SQ_LDS_BANK_CONFLICT", 0) / c, a.get("GRBM_GUI_ACTIVE", 0) / c
    print(f"{k[:62]:62s} {c:5d} {act:15.0f} {conf:14.0f} {100 * conf / max(act, 1):9.1f} {a.get('SQ_INSTS_LDS', 0) / c:10.0f} {gui:15.0f} {100 * act / 256 / max(gui, 1):10.1f} {a.get('SQ_INSTS_VALU', 0) / c:12.0f}")
PY
rm -rf $out/p1
