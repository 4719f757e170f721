python -m pytest tests/test_hip_parity.py tests/test_boundary_gpu.py -q -x -k "text_encoder or conformer or acoustic_train_step or layernorm or speech_predictor_end or two_rank_hip or textual or duration_trainer or soak or grouped" 2>&1 | grep -v "^$" | tail -6 > gpurun_out/t_fix.log
: > gpurun_out/phases.txt
run() { # name, env...
  n=$1; shift 1
  env "$@" STY_STEP_PROBE=1 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>gpurun_out/phase_err.txt | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],3), " ".join(f"{t:.2f}" for n,t in d.get("phases_ms",[])))
' $n >> gpurun_out/phases.txt 2>&1
}
run new X=1
run old STY_NO_LN_PARAM_SIDE=1
run new2 X=1
run old2 STY_NO_LN_PARAM_SIDE=1
echo done
