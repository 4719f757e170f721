python -m pytest tests/test_hip_parity.py -q -k "stem_weight_gradient" -s 2>&1 | tail -30 > gpurun_out/t_fix.log
