#!/bin/bash
# Kernel-trace summary of the c3 step with every internal side stream off (STY_NO_SIDE_STREAM / STY_NO_SE_STREAM):
# per-kernel durations without the stretch from sharing the chip -- the honest cost ranking of the step's kernels.
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/serial_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2
export STY_NO_SIDE_STREAM=1 STY_NO_SE_STREAM=1
B="python $R/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $O/trace -- $B --steps 6 --warmup 2 > $O/c3_serial_under_rocprof.json 2> $O/trace.log
cd $R
python tools/rocpd_summary.py $O/trace/*/*_results.db > $O/${tag}_c3_serial_kernel_stats.txt
rm -rf $O/trace
