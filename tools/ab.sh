#!/bin/bash
# run bench.py for each library variant given on the command line ("-" = product library); prints value / ms per step
wl=${WL:-c2}
for v in "$@"; do
  if [ "$v" == "-" ]; then unset STY_LIB_VARIANT; else export STY_LIB_VARIANT=$v; fi
  STY_PROF_SHAPES=1 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_${wl}_$v.json 2>/dev/null
  python - <<PY
import json
r=json.loads(open("gpurun_out/ab_${wl}_$v.json").read().strip().splitlines()[-1])
print("$v", "$wl", round(r["value"]), round(r["ms_per_step"],2))
PY
done
