#!/bin/bash
# A/B of environment switches / library variants on one GPU box:
#   WL=c3 tools/ab_env.sh "TAG1:VAR=1 VAR2=x" "TAG2:" "TAG3:STY_LIB_VARIANT=name" ...   (TAG: with nothing = product defaults)
wl=${WL:-c3}
steps=${STEPS:-20}
mkdir -p gpurun_out
for spec in "$@"; do
  tag=${spec%%:*}
  envs=${spec#*:}
  env $envs python bench.py --workload $wl --steps $steps --warmup 5 --no-extra --no-cpu-baseline --detail gpurun_out/ab_${wl}_${tag}.json > gpurun_out/ab_${wl}_${tag}.line 2> gpurun_out/ab_${wl}_${tag}.err
  python tools/ab_row.py gpurun_out/ab_${wl}_${tag}.json "$tag" "$envs"
done
