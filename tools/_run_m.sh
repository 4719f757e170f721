python -m pytest tests/test_hip_parity.py -q -k "prepare_train" 2>&1 | grep -v "^$" | tail -25 > gpurun_out/t_fix.log
echo done
