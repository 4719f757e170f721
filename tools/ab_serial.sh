# tuning aid: the serialised c3 step's kernel table (every kernel alone on the chip) under library variants; FILTER = grep -E pattern
O=$GRAFT_REPO_ROOT/gpurun_out/abser; mkdir -p $O; export TMPDIR=/tmp
for v in ${VARIANTS:-base}; do
  if [ $v == base ]; then unset STY_LIB_VARIANT; else export STY_LIB_VARIANT=$v; fi
  ( cd /tmp && STY_NO_SIDE_STREAM=1 STY_NO_SE_STREAM=1 rocprofv3 --kernel-trace --stats -d $O/tr_$v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > /dev/null 2> $O/tr_$v.log )
  python tools/rocpd_summary.py $O/tr_$v/*/*_results.db > $O/serial_$v.txt; rm -rf $O/tr_$v
  echo "== $v"; grep -E "${FILTER:-wgrad}" $O/serial_$v.txt | cut -c1-130
done
