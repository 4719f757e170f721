python tools/soak.py 300 > gpurun_out/soak.txt 2>&1
python - <<'PY' >> gpurun_out/soak.txt 2>&1
# 300 consecutive c3 steps (bf16 mode, train mode): time and finiteness
import os, time, json, subprocess, sys
r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extra", "--steps", "300", "--warmup", "5"], capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
print("c3 300 steps:", d["ms_per_step"], "ms/step", d["value"], "frames/s")
PY
echo done
