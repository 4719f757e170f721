#!/bin/bash
# Everything the round's profiles/ directory is made from (run on the GPU box through gpurun):
#   kernel-trace summaries (rocprofv3 --kernel-trace --stats) of the default bench command and of c5,
#   HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate passes) of the default bench command,
#   bench JSON lines of every workload.
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2   # as bench.py sets it for itself (rocprofv3 loads the HIP runtime first)
rocprofv3 --kernel-trace --stats -d $O/c2_trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/c2_under_rocprof.json 2> $O/c2_trace.log
rocprofv3 --kernel-trace --stats -d $O/c5_trace -- python $R/bench.py --workload c5 --steps 20 --warmup 5 --no-cpu-baseline > $O/c5_under_rocprof.json 2> $O/c5_trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c2_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/c2_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c2_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/c2_write.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c5_fetch -- python $R/bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/c5_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c5_write -- python $R/bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/c5_write.log
cd $R
python tools/pmc_traffic.py $O/c5_fetch $O/c5_write > $O/${tag}_c5_pmc_traffic.json
python tools/rocpd_summary.py $O/c2_trace/*/*_results.db > $O/${tag}_c2_train_kernel_stats.txt
python tools/rocpd_summary.py $O/c5_trace/*/*_results.db > $O/${tag}_c5_kernel_stats.txt
python tools/pmc_traffic.py $O/c2_fetch $O/c2_write > $O/${tag}_c2_pmc_traffic.json
python bench.py > $O/${tag}_bench_c2.json 2> $O/bench_c2.err
python bench.py --workload c2-fwd --no-cpu-baseline > $O/${tag}_bench_c2_fwd.json 2>/dev/null
python bench.py --workload c3-fp32 --steps 5 --warmup 2 --no-cpu-baseline > $O/${tag}_bench_c3_fp32.json 2>/dev/null
python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > $O/${tag}_bench_c3_bf16.json 2>/dev/null
python bench.py --workload c5-bf16 --steps 50 --warmup 10 --no-cpu-baseline > $O/${tag}_bench_c5_bf16.json 2>/dev/null
python bench.py --workload c5 --steps 50 --warmup 10 > $O/${tag}_bench_c5.json 2>/dev/null
python bench.py --workload tts --steps 20 --warmup 5 > $O/${tag}_bench_tts.json 2>/dev/null
rm -rf $O/c2_trace $O/c5_trace $O/c2_fetch/*/*agent_info.csv $O/c5_fetch/*/*agent_info.csv
ls -la $O
