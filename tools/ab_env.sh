#!/bin/bash
# A/B of environment switches on one GPU box: WL=c3 tools/ab_env.sh "TAG1:VAR=1 VAR2=x" "TAG2:" ...   (TAG: with nothing = product defaults)
wl=${WL:-c3}
steps=${STEPS:-20}
mkdir -p gpurun_out
for spec in "$@"; do
  tag=${spec%%:*}
  envs=${spec#*:}
  env $envs python bench.py --workload $wl --steps $steps --warmup 5 --no-extra --no-cpu-baseline --detail gpurun_out/ab_${wl}_${tag}.json > gpurun_out/ab_${wl}_${tag}.line 2> gpurun_out/ab_${wl}_${tag}.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/ab_${wl}_${tag}.line").read().strip().splitlines()[-1])
    print(f"${tag:20s} ${wl} {r['ms_per_step']:.2f} ms  serial {r.get('single_stream_step_ms', 0):.2f} ms   [{'${envs}'}]")
except Exception as e:
    print("${tag}", "FAILED", e)
PY
done
