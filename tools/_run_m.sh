python -m pytest tests/test_hip_parity.py tests/test_full_size.py -q -k "style or acoustic_train_step or twin or weight_gradient or grouped or full_size" 2>&1 | grep -v "^$" | tail -6 > gpurun_out/t_fix.log
: > gpurun_out/phases.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("c3", d["ms_per_step"], "serial", d["single_stream_step_ms"])
for k in d["single_stream_kernels"]:
    if "wgradb16" in k["name"] or "convp16" in k["name"]: print("  alone", k["name"], k["launches"], round(k["ms_per_step"],3), round(k["TFLOPs"],1))
for k in d["kernels"]:
    if "wgradb16" in k["name"] or "convp16" in k["name"]: print("  insitu", k["name"], k["launches"], round(k["ms_per_step"],3), round(k["TFLOPs"],1))
' >> gpurun_out/phases.txt 2>&1
done
python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 --workload c2 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("c2", d["ms_per_step"])' >> gpurun_out/phases.txt
echo done
