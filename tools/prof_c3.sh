#!/bin/bash
# quick kernel-trace summary of the default bench command (c3): gpurun_out/prof_c3.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2
rm -rf /tmp/prof_q
rocprofv3 --kernel-trace --stats -d /tmp/prof_q -- python $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 ${@} > /tmp/prof_q.json 2> /tmp/prof_q.log
python $R/tools/rocpd_summary.py $(find /tmp/prof_q -name "*_results.db" | head -1)
