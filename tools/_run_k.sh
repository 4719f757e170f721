# round 4, FFT front end: parity of the front-end tests, then A/B of the c3 / c2 step against the GEMM front end
python -m pytest tests/test_hip_parity.py -q -s -x -k "fft_front_end or mel_front_end or multi_spectrogram or acoustic_losses_forward_backward or acoustic_step_forward or acoustic_train_step_gradients" 2>&1 | grep -v "^$" | tail -80 > gpurun_out/t_fft.log
B="python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("single_stream_step_ms"))'
: > gpurun_out/ab.txt
$B 2>gpurun_out/err_c3.txt | python -c "$P" c3_fft >> gpurun_out/ab.txt 2>&1
STY_DFT_GEMM=1 $B 2>/dev/null | python -c "$P" c3_gemm >> gpurun_out/ab.txt 2>&1
STY_FFT_TF=8 $B 2>/dev/null | python -c "$P" c3_fft_tf8 >> gpurun_out/ab.txt 2>&1
$B --workload c2 2>/dev/null | python -c "$P" c2_fft >> gpurun_out/ab.txt 2>&1
STY_DFT_GEMM=1 $B --workload c2 2>/dev/null | python -c "$P" c2_gemm >> gpurun_out/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fft -- python $R/bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 > /dev/null 2> $R/gpurun_out/prof_fft.log
cd $R
python tools/rocpd_summary.py gpurun_out/prof_fft/*/*_results.db > gpurun_out/fft_kernel_stats.txt 2>&1 || true
rm -rf gpurun_out/prof_fft
echo done
