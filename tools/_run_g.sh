python -m pytest tests/test_hip_parity.py -x -q -s -k "operand_twins" 2>&1 | tail -40 > gpurun_out/t_twins.log
python -m pytest tests/test_hip_parity.py -x -q -k "style or taps or acoustic_train or persistent" 2>&1 | tail -15 > gpurun_out/t_reg.log
B="python bench.py --no-cpu-baseline --no-extra --steps 12 --warmup 3"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('twins', d['ms_per_step'], d['single_stream_step_ms']); [print(k) for k in d['single_stream_kernels'] if 'wgradb' in k['name'] or 'convp16' in k['name'] or 'twin' in k['name']]" > gpurun_out/sweep_tw.txt 2>&1
STY_NO_TWINS=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no twins', d['ms_per_step'], d['single_stream_step_ms'])" >> gpurun_out/sweep_tw.txt 2>&1
for f in 0.25 0.125; do STY_WG_PARTIAL_FRAC=$f $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('twins frac $f', d['ms_per_step'], d['single_stream_step_ms'])"; done >> gpurun_out/sweep_tw.txt 2>&1
python -m pytest tests/test_full_size.py -x -q -s -k "c3_train_step_full" 2>&1 | tail -150 > gpurun_out/t_c3full.log
echo done
