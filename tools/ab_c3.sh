# tuning aid: c3 step time under library variants (tools/build_variant.sh), alternating;  VARIANTS="base x base x" tools/ab_c3.sh
O=$GRAFT_REPO_ROOT/gpurun_out/abc3; mkdir -p $O
for v in ${VARIANTS:-base}; do
  if [ $v == base ]; then unset STY_LIB_VARIANT; else export STY_LIB_VARIANT=$v; fi
  python bench.py --no-cpu-baseline --no-extra --workload ${WL:-c3} --steps ${STEPS:-20} --warmup 5 2>/dev/null | tail -1 > $O/line.json
  python -c "import json;d=json.load(open('$O/line.json'));print('$v', d['ms_per_step'], 'single-stream', d.get('single_stream_step_ms'))"
done
