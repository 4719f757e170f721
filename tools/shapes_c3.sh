STY_PROF_SHAPES=1 STY_NO_SIDE_STREAM=1 STY_NO_SE_STREAM=1 python bench.py --workload c3 --no-cpu-baseline --no-extra --steps 5 --warmup 2 --detail gpurun_out/shapes_c3.json > /dev/null 2> gpurun_out/shapes.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/shapes_c3.json'))
with open('gpurun_out/shapes_c3.txt','w') as f:
    f.write("serial step %s\n"%d["ms_per_step"])
    for k in d["kernels"]:
        f.write("%8.3f ms %4d %8.1f us %7.1f TF %7.0f GB/s  %s\n" % (k["ms_per_step"], k["launches"], 1e3*k["ms_per_step"]/k["launches"], k["TFLOPs"], k["GBps"], k["name"]))
PY
head -5 gpurun_out/shapes_c3.txt
