#!/bin/bash
# usage (on the GPU box): tools/pmc_layer.sh "64,64,3,80" outdir   -> per-kernel PMC summary of one conv layer
# Counter passes are separate rocprofv3 runs (never combined with trace domains).
only=$1; out=${2:-gpurun_out/pmc_layer}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for pm in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
          "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_INSTS_SMEM" \
          "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pm --output-format csv -d $root/$out/p$i -- python3 $root/tools/bench_conv.py --only $only --iters 2 ${EXTRA_ARGS} > /dev/null 2>&1
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv" not in k and "stem" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])
with open(f"{out}/summary.txt", "w") as fo:
    for k, cs in agg.items():
        print(k, "grid,wg,lds,vgpr,agpr=", meta[k], file=fo)
        for c in sorted(cs):
            v = cs[c]; print(f"   {c:40s} {sum(v)/len(v):16.1f}  (n={len(v)})", file=fo)
print(open(f"{out}/summary.txt").read())
PY
