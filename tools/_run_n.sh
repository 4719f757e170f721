python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 > gpurun_out/t_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
