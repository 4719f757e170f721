#!/bin/bash
# PMC breakdown of conv32p_kernel on one shape (tools/conv32p_bench.py <shape> <mode> plain), separate passes per counter set
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/conv32p_pmc
rm -rf $out; mkdir -p $out
shape=${1:-0}; mode=${2:-0}
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python $GRAFT_REPO_ROOT/tools/conv32p_bench.py $shape $mode plain > $out/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("gpurun_out/conv32p_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv32p" not in k: continue
        agg[k[:60]][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k[:60],r["Counter_Name"])]+=1
for key,v in agg.items():
    print(key)
    for c,val in sorted(v.items()): print(f"    {c:32s} {val/cnt[(key,c)]:18.1f}")
PY
