#!/bin/bash
# kernel-trace summary of the c5-bf16 forward (tuning aid)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5b; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=2
rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-extra --workload c5-bf16 --steps 20 --warmup 5 > $O/line.json 2> $O/trace.log
cd $R; python tools/rocpd_summary.py $O/trace/*/*_results.db > $O/c5b_kernel_stats.txt; rm -rf $O/trace
head -24 $O/c5b_kernel_stats.txt | cut -c1-150
