python -m pytest tests/test_hip_parity.py -q -k "attention_bf16 or attention_backward" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/t_attn.log
run() { # name, env...
  n=$1; shift
  env "$@" STY_STEP_PROBE=1 python bench.py --no-cpu-baseline --no-extra --steps 8 --warmup 3 2>gpurun_out/phase_err.txt | python -c '
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"])
for n,t in d.get("phases_ms",[]): print(f"{t:9.3f}  {n}")
' $n >> gpurun_out/phases.txt 2>&1
}
: > gpurun_out/phases.txt
run new X=1
run no_attn16 STY_NO_ATTN16=1
run new2 X=1
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_a -- python $R/bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 > /dev/null 2> $R/gpurun_out/prof_a.log
cd $R
python tools/rocpd_summary.py gpurun_out/prof_a/*/*_results.db > gpurun_out/a_kernel_stats.txt 2>&1 || true
rm -rf gpurun_out/prof_a
echo done
