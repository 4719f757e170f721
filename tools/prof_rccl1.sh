#!/bin/bash
# Where the step under RCCL (world size 1, every gradient bucket all-reduced) loses time against the step without a process group:
# kernel trace of both, RCCL's kernels and the streams they run on, and the step under GPU_MAX_HW_QUEUES = 2 / 3 / 4.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_rccl1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2
rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --rccl1 --no-extra --no-cpu-baseline --steps 6 --warmup 2 > $O/rccl1_under_rocprof.json 2> $O/trace.log
cd $R
python tools/rocpd_summary.py $O/trace/*/*_results.db > $O/rccl1_kernel_stats.txt
python tools/stream_busy.py $O/trace/*/*_results.db 6 > $O/rccl1_streams.txt
python tools/step_gaps.py $O/trace/*/*_results.db 6 150 > $O/rccl1_gaps.txt
rm -rf $O/trace
for q in 2 3 4; do
  for mode in "" "--rccl1"; do
    GPU_MAX_HW_QUEUES=$q python bench.py $mode --no-extra --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GPU_MAX_HW_QUEUES=$q', '$mode' or 'no process group', round(d['ms_per_step'], 2), 'ms')"
  done
done | tee $O/rccl1_queues.txt
