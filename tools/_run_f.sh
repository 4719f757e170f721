python -m pytest tests/test_hip_parity.py -x -q -s -k "bf16_operand_twins" 2>&1 | tail -40 > gpurun_out/t_twins.log
python -m pytest tests/test_hip_parity.py -x -q -k "style or dense_conv1d or persistent_conv16 or acoustic_train" 2>&1 | tail -15 > gpurun_out/t_reg.log
B="python bench.py --no-cpu-baseline --no-extra --steps 12 --warmup 3"
for f in 0 0.5 0.25; do STY_WG_PARTIAL_FRAC=$f $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frac $f', d['ms_per_step'], d['single_stream_step_ms'])"; done > gpurun_out/sweep_frac.txt 2>&1
STY_NO_TWINS=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no twins', d['ms_per_step'], d['single_stream_step_ms']); [print(k) for k in d['single_stream_kernels'] if 'wgradb' in k['name'] or 'convp16' in k['name'] or 'twin' in k['name']]" >> gpurun_out/sweep_frac.txt 2>&1
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('twins', d['ms_per_step'], d['single_stream_step_ms']); [print(k) for k in d['single_stream_kernels'] if 'wgradb' in k['name'] or 'convp16' in k['name'] or 'twin' in k['name']]" >> gpurun_out/sweep_frac.txt 2>&1
echo done
