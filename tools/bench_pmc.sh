#!/bin/bash
# PMC breakdown (SQ wave-cycle buckets, instruction mix) of selected kernels inside a bench run
#   tools/bench_pmc.sh <workload> <kernel-name-substring> [more substrings...]
wl=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/bench_pmc
rm -rf $out; mkdir -p $out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$@" <<'PY'
import csv, glob, collections, sys
subs=sys.argv[1:]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("gpurun_out/bench_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not any(s in k for s in subs): continue
        key=k[:70]
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(key,r["Counter_Name"])]+=1
for key,v in agg.items():
    print(key)
    for c,val in sorted(v.items()): print(f"    {c:32s} {val/cnt[(key,c)]:16.1f}")
PY
