# tuning aid: kernel rows of the fused ConvNeXt32 forward under library variants (tools/build_variant.sh), c5-bf16 and c3
O=$GRAFT_REPO_ROOT/gpurun_out/slp; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=2
for v in ${VARIANTS:-base}; do
  if [ $v == base ]; then unset STY_LIB_VARIANT; else export STY_LIB_VARIANT=$v; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/tr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --workload ${WL:-c5-bf16} --steps 20 --warmup 5 > $O/line_$v.json 2> $O/log_$v.txt )
  python tools/rocpd_summary.py $O/tr/*/*_results.db > $O/c5b_$v.txt; rm -rf $O/tr
  python bench.py --no-cpu-baseline --no-extra --workload ${WL:-c5-bf16} --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/c5b_$v.json
  echo "== $v"; grep -h -E "convnext32" $O/c5b_$v.txt | cut -c1-120
  python -c "import json;print('c5b',json.load(open('$O/c5b_$v.json'))['ms_per_step'])"
done
