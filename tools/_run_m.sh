python -m pytest tests/test_hip_parity.py tests/test_boundary_gpu.py -q -k "bench_two_ranks or bench_gpus_flag or two_rank or rccl" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/t_fix.log
echo done
