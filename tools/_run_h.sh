python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/t_gpu.log
bash tools/profile_round.sh r04 > gpurun_out/profile_round.log 2>&1
echo done
