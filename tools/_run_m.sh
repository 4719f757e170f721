python -m pytest tests/test_hip_parity.py -q -x -k "attention_backward or speech_predictor or acoustic_train_step or textual or duration" 2>&1 | grep -v "^$" | tail -15 > gpurun_out/t_fix.log
run() { # name, env...
  n=$1; shift
  env "$@" STY_STEP_PROBE=1 python bench.py --no-cpu-baseline --no-extra --steps 8 --warmup 3 2>gpurun_out/phase_err.txt | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"])
for n,t in d.get("phases_ms",[]): print(f"{t:9.3f}  {n}")
' $n >> gpurun_out/phases.txt 2>&1
}
: > gpurun_out/phases.txt
run new X=1
run old_attn STY_NO_ATTN_BWD_SMALL=1
run new2 X=1
echo done
