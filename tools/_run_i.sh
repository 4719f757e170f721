B="python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("single_stream_step_ms"))'
STY_PROF_SHAPES=1 STY_NO_SIDE_STREAM=1 STY_NO_SE_STREAM=1 $B 2>/dev/null | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("serial step", d["ms_per_step"])
for k in d["kernels"]:
    print("%8.3f ms %4d  %7.1f TF %7.0f GB/s  %s" % (k["ms_per_step"], k["launches"], k["TFLOPs"], k["GBps"], k["name"]))
' > gpurun_out/shapes_c3.txt 2>&1
$B 2>/dev/null | python -c "$P" base > gpurun_out/ab.txt 2>&1
STY_NO_TWINS=1 $B 2>/dev/null | python -c "$P" no_twins >> gpurun_out/ab.txt 2>&1
STY_T128_TILES=256 $B 2>/dev/null | python -c "$P" t128_256 >> gpurun_out/ab.txt 2>&1
STY_KS2_WGS=100000 $B 2>/dev/null | python -c "$P" ks2_all >> gpurun_out/ab.txt 2>&1
STY_T64_TILES=128 $B 2>/dev/null | python -c "$P" t64_128 >> gpurun_out/ab.txt 2>&1
for v in 384 512; do STY_WG_TARGET=$v $B 2>/dev/null | python -c "$P" wg_target_$v >> gpurun_out/ab.txt 2>&1; done
for v in 0.5 0.25; do STY_WG_PARTIAL_FRAC=$v $B 2>/dev/null | python -c "$P" wg_frac_$v >> gpurun_out/ab.txt 2>&1; done
$B 2>/dev/null | python -c "$P" base2 >> gpurun_out/ab.txt 2>&1
echo done
