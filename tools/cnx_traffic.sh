#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one --pmc pass each) of the 75T-rate training kernels on tools/cnx_pmc.py's workload
# (one ConvNeXt32 block + one resblock, B = 8, T = 39 000: 312 000 positions, x = 39.9 MB as fp32) -> gpurun_out/cnx_traffic.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/cnx_traffic; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python $R/tools/cnx_pmc.py 3 > $out/$c.log 2>&1
done
cd $R
python tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE > $out/traffic.json
python - <<'PY' > gpurun_out/cnx_traffic.txt
import json
d = json.load(open("gpurun_out/cnx_traffic/traffic.json"))
print("# tools/cnx_traffic.sh: bytes per launch, B = 8, T = 39000 (312 000 positions; a [B][32][T] fp32 tensor = 39.9 MB)")
for k, v in sorted(d.items(), key=lambda kv: -(kv[1].get("fetch_bytes_per_launch", 0) + kv[1].get("write_bytes_per_launch", 0))):
    if "launches" not in v: continue
    print("%8.1f MB fetched %8.1f MB written  x%-3d %s" % (v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["launches"], k[:110]))
PY
cat gpurun_out/cnx_traffic.txt | head -40
