python -m pytest tests/test_hip_parity.py -q -k "fft_front_end or acoustic_losses_forward_backward or acoustic_train_step_gradients or spectrogram_discriminators" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/t_fix.log
python -m pytest tests/test_discriminators.py tests/test_boundary_gpu.py -q 2>&1 | grep -v "^$" | tail -10 >> gpurun_out/t_fix.log
B="python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("single_stream_step_ms"))'
: > gpurun_out/ab.txt
$B 2>gpurun_out/err_c3.txt | python -c "$P" c3_span >> gpurun_out/ab.txt 2>&1
STY_FFT_NO_SPAN=1 $B 2>/dev/null | python -c "$P" c3_nospan >> gpurun_out/ab.txt 2>&1
$B 2>/dev/null | python -c "$P" c3_span_again >> gpurun_out/ab.txt 2>&1
STY_PROF_SHAPES=1 python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 2 2>/dev/null > gpurun_out/c3_shapes.json
echo done
