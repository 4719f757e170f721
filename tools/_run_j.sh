python -m pytest tests/test_hip_parity.py -q -s -k "operand_twins_equal or grouped_weight_gradient_reduction or bf16_weight_gradient_kernels_match" 2>&1 | grep -v "^$" | tail -150 > gpurun_out/t_fail5.log
python -m pytest tests/test_full_size.py -q -k "c3_train_step_full" 2>&1 | tail -5 > gpurun_out/t_c3full.log
bash tools/_run_i.sh
B="python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("single_stream_step_ms"))'
STY_NO_DEFERRED_REDUCE=1 $B 2>/dev/null | python -c "$P" c3_no_deferred >> gpurun_out/ab.txt 2>&1
$B --workload c2 2>/dev/null | python -c "$P" c2_base >> gpurun_out/ab.txt 2>&1
STY_NO_DEFERRED_REDUCE=1 $B --workload c2 2>/dev/null | python -c "$P" c2_no_deferred >> gpurun_out/ab.txt 2>&1
$B --workload c2 2>/dev/null | python -c "$P" c2_base2 >> gpurun_out/ab.txt 2>&1
$B --workload c3-gan --steps 5 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("c3-gan", d["ms_per_step"], d["host_issue_ms_per_step"])' >> gpurun_out/ab.txt 2>&1
echo done
