#!/bin/bash
# LDS bank-conflict cycles per LDS instruction for every kernel of a workload (one --pmc pass) -> gpurun_out/lds_conflicts_<wl>.txt
wl=${1:-c3}
cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/ldsc; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $out/p -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/log.txt 2>&1
cd $R
python - $wl <<'PY' > gpurun_out/lds_conflicts_$wl.txt
import csv, glob, collections, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/ldsc/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void sty::", "").replace("sty::", "").replace("(anonymous namespace)::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
rows = []
for k, v in agg.items():
    a = {c: val / n[(k, c)] for c, val in v.items()}
    li = a.get("SQ_INSTS_LDS", 0)
    if li <= 0: continue
    rows.append((a.get("SQ_LDS_BANK_CONFLICT", 0) * n[(k, "SQ_INSTS_LDS")], k, a.get("SQ_LDS_BANK_CONFLICT", 0) / li, a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1), li / max(a.get("SQ_WAVES", 1), 1), n[(k, "SQ_INSTS_LDS")]))
rows.sort(reverse=True)
print("# tools/lds_conflicts.sh %s: per kernel -- conflict cycles per LDS instruction | conflict / active LDS cycles | LDS instructions per wave | launches" % sys.argv[1])
for tot, k, per, frac, lpw, cnt in rows[:45]:
    print("%7.2f  %5.2f  %8.0f  x%-4d %s" % (per, frac, lpw, cnt, k[:100]))
PY
head -40 gpurun_out/lds_conflicts_$wl.txt
