python -m pytest tests/test_boundary_gpu.py tests/test_hip_parity.py -q -k "style or acoustic_train_step or twin or weight_gradient or grouped or full_size" 2>&1 | grep -v "^$" | tail -6 > gpurun_out/t_fix.log
run() { # name, wl, env...
  n=$1; wl=$2; shift 2
  env "$@" STY_STEP_PROBE=1 python bench.py --no-cpu-baseline --no-extra --steps 8 --warmup 3 --workload $wl 2>gpurun_out/phase_err.txt | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"])
for n,t in d.get("phases_ms",[]): print(f"{t:9.3f}  {n}")
' $n >> gpurun_out/phases.txt 2>&1
}
: > gpurun_out/phases.txt
run c3 c3 X=1
run c3_dual0 c3 STY_SIDE_DUAL=0
run c3_dual2 c3 STY_SIDE_DUAL=2
run c3b c3 X=1
run c3_dual0b c3 STY_SIDE_DUAL=0
run c2 c2 X=1
run c2_dual0 c2 STY_SIDE_DUAL=0
run c2_dual2 c2 STY_SIDE_DUAL=2
echo done
