# FFT v2 (radix-4, float4 loads, 1024 threads) + dwconv2d_s2_bwd with 4 chunks per workgroup
python -m pytest tests/test_hip_parity.py -q -s -x -k "fft_front_end or mel_front_end or multi_spectrogram or acoustic_losses_forward_backward or style_encoder or acoustic_train_step_gradients or multi_stream_step" 2>&1 | grep -v "^$" | tail -60 > gpurun_out/t_fft.log
B="python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("single_stream_step_ms"))'
: > gpurun_out/ab.txt
$B 2>gpurun_out/err_c3.txt | python -c "$P" c3_fft2 >> gpurun_out/ab.txt 2>&1
STY_FB_GEMM=1 $B 2>/dev/null | python -c "$P" c3_fbgemm >> gpurun_out/ab.txt 2>&1
$B 2>/dev/null | python -c "$P" c3_fft2_again >> gpurun_out/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fft -- python $R/bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 > /dev/null 2> $R/gpurun_out/prof_fft.log
cd $R
python tools/rocpd_summary.py gpurun_out/prof_fft/*/*_results.db > gpurun_out/fft_kernel_stats.txt 2>&1 || true
python tools/stream_busy.py gpurun_out/prof_fft/*/*_results.db 6 > gpurun_out/fft_c3_streams.txt 2>&1 || true
python tools/step_gaps.py gpurun_out/prof_fft/*/*_results.db 6 150 > gpurun_out/fft_c3_gaps.txt 2>&1 || true
rm -rf gpurun_out/prof_fft
echo done
