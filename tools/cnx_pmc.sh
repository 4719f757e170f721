#!/bin/bash
# SQ counters of the 75T-rate training kernels (tools/cnx_pmc.py; `tools/cnx_pmc.sh se`: the style encoder's, tools/se_pmc.py),
# one --pmc pass per counter set; summary -> gpurun_out/cnx_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
wl=${1:-cnx}
export PMC_FILTER="convnext32 wgrad_cnx conv32p pro_bwd wgradp32 dwconv"
[ $wl = se ] && export PMC_FILTER="convq convp16 wgradb16 wgradb_ dwconv2d avgpool stem pool_fc twin_cast"
out=$R/gpurun_out/cnx_pmc
rm -rf $out; mkdir -p $out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python $R/tools/${wl}_pmc.py 3 > $out/$tag.log 2>&1
done
cd $R
python - > gpurun_out/cnx_pmc.txt <<'PY'
import csv, glob, collections, re, os
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/cnx_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void sty::", "").replace("sty::", "")
        if not any(s in k for s in os.environ["PMC_FILTER"].split()): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
print("# tools/cnx_pmc.sh: SQ counters per launch (mean over launches), B = 8, T = 39000, bf16 mode")
for k, v in sorted(agg.items()):
    a = {c: val / cnt[(k, c)] for c, val in v.items()}
    w = max(a.get("SQ_WAVES", 0), 1)
    wc = max(a.get("SQ_WAVE_CYCLES", 0), 1)
    print(k)
    print("   waves %d  per wave: VALU %.0f (trans %.0f)  SALU %.0f  LDS %.0f  VMEM rd %.0f wr %.0f  wave-cycles %.0f" % (
        w, a.get("SQ_INSTS_VALU", 0) / w, a.get("SQ_INSTS_VALU_TRANS", 0) / w, a.get("SQ_INSTS_SALU", 0) / w, a.get("SQ_INSTS_LDS", 0) / w,
        a.get("SQ_INSTS_VMEM_RD", 0) / w, a.get("SQ_INSTS_VMEM_WR", 0) / w, wc / w))
    print("   of the wave cycles: waiting (any) %.2f  waiting on an instruction %.2f  issuing %.2f | active VALU %.2f SALU %.2f VMEM %.2f LDS %.2f  wait LDS %.2f" % (
        a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        a.get("SQ_ACTIVE_INST_SCA", 0) / wc, a.get("SQ_ACTIVE_INST_VMEM", 0) / wc, a.get("SQ_ACTIVE_INST_LDS", 0) / wc, a.get("SQ_WAIT_INST_LDS", 0) / wc))
    print("   MFMA busy / GUI active %.3f   LDS bank conflict cycles %.0f" % (a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(a.get("GRBM_GUI_ACTIVE", 1), 1) / 1024 * 1, a.get("SQ_LDS_BANK_CONFLICT", 0)))
PY
cat gpurun_out/cnx_pmc.txt
