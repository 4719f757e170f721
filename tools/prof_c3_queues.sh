# the c3 step's stream picture under different numbers of hardware queues: gpurun_out/c3_{gaps,streams}_q<N>.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for Q in ${@:-2 4 8}; do
  export GPU_MAX_HW_QUEUES=$Q
  rm -rf /tmp/prof_q
  rocprofv3 --kernel-trace --stats -d /tmp/prof_q -- python $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > /tmp/prof_q.json 2> /tmp/prof_q.log
  DB=$(find /tmp/prof_q -name "*_results.db" | head -1)
  python $R/tools/step_gaps.py $DB 6 150 > $R/gpurun_out/c3_gaps_q$Q.txt
  python $R/tools/stream_busy.py $DB 6 > $R/gpurun_out/c3_streams_q$Q.txt
  python - $DB <<'PY' >> $R/gpurun_out/c3_streams_q$Q.txt
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("columns:", cols)
if "stream_id" in cols and "queue_id" in cols:
    for r in c.execute("select stream_id, queue_id, count(*) from kernels group by stream_id, queue_id"):
        print("stream", r[0], "queue", r[1], "kernels", r[2])
PY
  tail -1 /tmp/prof_q.json | python -c "import sys,json; print('Q$Q ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])" >> $R/gpurun_out/c3_streams_q$Q.txt
done
