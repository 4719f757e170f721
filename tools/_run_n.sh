python -m pytest tests/test_boundary_gpu.py -q -k "two_rank_hip" 2>&1 | tail -80 > gpurun_out/t_fix.log
