R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2
rm -rf /tmp/prof_q
rocprofv3 --kernel-trace --stats -d /tmp/prof_q -- python $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > /tmp/prof_q.json 2> /tmp/prof_q.log
DB=$(find /tmp/prof_q -name "*_results.db" | head -1)
python $R/tools/se_fwd_list.py $DB > $R/gpurun_out/se_fwd_list.txt
