#!/bin/bash
# per-(kernel family, problem shape) table of the serial c3 step: tools/shapes_c3.sh [tag] (environment passes through)
tag=${1:-c3}
STY_PROF_SHAPES=1 STY_NO_SIDE_STREAM=1 STY_NO_SE_STREAM=1 python bench.py --workload c3 --no-cpu-baseline --no-extra --steps 5 --warmup 2 --detail gpurun_out/shapes_$tag.json > /dev/null 2> gpurun_out/shapes_$tag.err
python - $tag <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open(f'gpurun_out/shapes_{tag}.json'))
with open(f'gpurun_out/shapes_{tag}.txt', 'w') as f:
    f.write("serial step %s\n" % d["ms_per_step"])
    for k in d["kernels"]:
        f.write("%8.3f ms %4d %8.1f us %7.1f TF %7.0f GB/s  %s\n" % (k["ms_per_step"], k["launches"], 1e3 * k["ms_per_step"] / k["launches"], k["TFLOPs"], k["GBps"], k["name"]))
PY
