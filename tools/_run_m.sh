python -m pytest tests/test_hip_parity.py -q -k "operand_twins_equal or bf16_weight_gradient_kernels_match or bench_two_ranks" 2>&1 | grep -v "^$" | tail -150 > gpurun_out/t_fix.log
echo done
