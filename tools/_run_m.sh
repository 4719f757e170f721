: > gpurun_out/phases.txt
python - <<'PY' >> gpurun_out/phases.txt 2>&1
import ctypes as C
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
print("prio range rc", hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "least", lo.value, "greatest", hi.value)
PY
run() { # name, env...
  n=$1; shift 1
  env "$@" STY_STEP_PROBE=1 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>gpurun_out/phase_err.txt | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],3), " ".join(f"{t:.2f}" for n,t in d.get("phases_ms",[])))
' $n >> gpurun_out/phases.txt 2>&1
}
run base X=1
run prio1 STY_SE_STREAM_PRIO=1
run prio0 STY_SE_STREAM_PRIO=0
run base2 X=1
run prio1b STY_SE_STREAM_PRIO=1
run prio2 STY_SE_STREAM_PRIO=2
echo done
