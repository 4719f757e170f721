: > gpurun_out/phases.txt
run() { # name, env...
  n=$1; shift 1
  env "$@" STY_STEP_PROBE=1 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>gpurun_out/phase_err.txt | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],3), " ".join(f"{t:.2f}" for n,t in d.get("phases_ms",[])))
' $n >> gpurun_out/phases.txt 2>&1
}
run base X=1
run wg1024 STY_WG_TARGET=1024
run wg896 STY_WG_TARGET=896
run wg640 STY_WG_TARGET=640
run base2 X=1
run wg1024b STY_WG_TARGET=1024
run free16 STY_CONVP16_FREE_CUS=16
run free48 STY_CONVP16_FREE_CUS=48
echo done
