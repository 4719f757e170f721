#!/bin/bash
# one traced c3 run -> per-stream busy summary, main-stream gaps, and the kernel timeline (CSV) of one step
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2
rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > $O/line.json 2> $O/trace.log
cd $R
db=$(ls $O/trace/*/*_results.db | head -1)
python tools/stream_busy.py $db 6 > $O/streams.txt
python tools/step_gaps.py $db 6 60 > $O/gaps.txt
python tools/timeline.py $db 2>/dev/null > $O/timeline_all.csv
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/tl/timeline_all.csv')))[1:]
# the last complete timed step before the single-stream extras: take the window between the 6th- and 5th-from-last groups of adamw launches
ad=[i for i,r in enumerate(rows) if 'adamw_kernel' in r[4]]
# group consecutive adamw launches
groups=[]
for i in ad:
    t=float(rows[i][0])
    if groups and t-groups[-1][-1][0] < 20000: groups[-1].append((t,i))
    else: groups.append([(t,i)])
ends=[g[-1] for g in groups]
print(len(groups),'adamw groups')
k=len(ends)-7
a,b=ends[k][1]+1, ends[k+1][1]+1
t0=float(rows[a][0])
with open('gpurun_out/tl/step.csv','w') as f:
    for r in rows[a:b]:
        f.write("%.1f,%s,%s,%s\n"%(float(r[0])-t0, r[1], r[3], r[4][:70]))
print('step rows',b-a,'window ms',(float(rows[b-1][0])-t0)/1e3)
PY
rm -rf $O/trace $O/timeline_all.csv
tail -3 $O/streams.txt
