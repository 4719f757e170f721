# tuning aid: c3 step time + phase stamps under environment settings;  tools/ab_env_phases.sh "A=1" "A=2 B=3" ...  ("-" = defaults)
for e in "$@"; do
  if [ "$e" == "-" ]; then e=""; fi
  env $e STY_STEP_PROBE=1 python bench.py --no-cpu-baseline --no-extra --steps ${STEPS:-20} --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
p={n.split(' (')[0]:t for n,t in d.get('phases_ms',[])}
print('%-40s step %.3f  d_style %.2f  style bwd done %.2f  predictor bwd done %.2f' % (sys.argv[1] or 'defaults', d['ms_per_step'], p.get('d_style ready',0), p.get('style encoder backward done',0), p.get('predictor backward done',0)))
" "$e"
done
