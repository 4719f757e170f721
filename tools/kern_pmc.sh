#!/bin/bash
# PMC breakdown of the 75T-rate building blocks (tools/prof_kernels.py) on the GPU box
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/kern_pmc
rm -rf $out; mkdir -p $out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py 3 > $out/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("gpurun_out/kern_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv1d_mfma" not in k and "convnext32" not in k: continue
        key=(k[:64], r.get("Grid_Size"))
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(key,r["Counter_Name"])]+=1
for key,v in agg.items():
    print(key)
    for c,val in sorted(v.items()): print(f"    {c:32s} {val/cnt[(key,c)]:16.1f}")
PY
