#!/bin/bash
# Instruction budget of ONE inference step (bench.py --serial) per kernel family from rocprofv3 PMC counters: where the
# chip's VALU / SALU / MFMA issue slots go.  Counter passes are separate runs, never combined with trace domains.
# usage (GPU box): tools/pmc_step.sh [outdir]   -> <outdir>/summary.txt
out=${1:-gpurun_out/pmc_step}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for pm in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
          "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
          "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $pm --output-format csv -d $root/$out/p$i -- python3 $root/bench.py --serial --steps 2 --warmup 1 --input-batches 1 --no-cpu-baseline --no-kernel-profile > /dev/null 2>&1
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
steps = 3.0 + 0  # warmup 1 + steps 2 replays (+ the two eager walks of compile() -> see calls column)
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_VALU_MFMA_BUSY_CYCLES",
         "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE",
         "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS"]
with open(f"{out}/summary.txt", "w") as fo:
    print("kernel".ljust(60), "calls", " ".join(n.replace("SQ_", "")[:14].rjust(15) for n in names), file=fo)
    tot = collections.defaultdict(float)
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", 0)):
        c = cnt[k].get("SQ_INSTS_VALU", 1)
        print(k[:60].ljust(60), f"{c:5d}", " ".join(f"{agg[k].get(n, 0):15.0f}" for n in names), file=fo)
        for n in names: tot[n] += agg[k].get(n, 0)
    print("TOTAL".ljust(60), "     ", " ".join(f"{tot[n]:15.0f}" for n in names), file=fo)
print(open(f"{out}/summary.txt").read())
PY
