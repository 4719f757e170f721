#!/bin/bash
# Everything the round's profiles/ directory is made from (run on the GPU box through gpurun):
#   kernel-trace summaries (rocprofv3 --kernel-trace --stats) of the default bench command (c3) and of c5 / c2,
#   HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate passes) of the default bench command and of c5,
#   the per-stream picture of one c3 step, bench JSON lines of every workload.
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2   # as bench.py sets it for itself (rocprofv3 loads the HIP runtime first)
B="python $R/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats -d $O/c3_trace -- $B --steps 6 --warmup 2 > $O/c3_under_rocprof.json 2> $O/c3_trace.log
rocprofv3 --kernel-trace --stats -d $O/c5_trace -- $B --workload c5 --steps 20 --warmup 5 > $O/c5_under_rocprof.json 2> $O/c5_trace.log
rocprofv3 --kernel-trace --stats -d $O/c2_trace -- $B --workload c2 --steps 10 --warmup 3 > $O/c2_under_rocprof.json 2> $O/c2_trace.log
rocprofv3 --kernel-trace --stats -d $O/c5b_trace -- $B --workload c5-bf16 --steps 20 --warmup 5 > $O/c5_bf16_under_rocprof.json 2> $O/c5b_trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c3_fetch -- $B --steps 2 --warmup 1 > /dev/null 2> $O/c3_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c3_write -- $B --steps 2 --warmup 1 > /dev/null 2> $O/c3_write.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c5_fetch -- $B --workload c5 --steps 5 --warmup 2 > /dev/null 2> $O/c5_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c5_write -- $B --workload c5 --steps 5 --warmup 2 > /dev/null 2> $O/c5_write.log
# ... and of the other workloads with a roofline of their own (c2; c5-bf16: the tracked vocoder figure since round 5), on THIS round's library
for wl in c2 c5-bf16; do
  n=2; [ $wl = c2 ] && n=4
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${wl}_fetch -- $B --workload $wl --steps $n --warmup 1 > /dev/null 2> $O/${wl}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${wl}_write -- $B --workload $wl --steps $n --warmup 1 > /dev/null 2> $O/${wl}_write.log
done
cd $R
for wl in c2 c5-bf16; do
  np=0; [ $wl = c5-bf16 ] && np=4   # c5-bf16: 2 timed + 1 warm-up + 1 profiled forward in the command above
  python tools/pmc_traffic.py $O/${wl}_fetch $O/${wl}_write $np > $O/${tag}_${wl}_pmc_traffic.json
  cp $O/${tag}_${wl}_pmc_traffic.json $R/profiles/
  rm -rf $O/${wl}_fetch $O/${wl}_write
done
python tools/pmc_traffic.py $O/c3_fetch $O/c3_write > $O/${tag}_c3_pmc_traffic.json
python tools/pmc_traffic.py $O/c5_fetch $O/c5_write 8 > $O/${tag}_c5_pmc_traffic.json   # 5 timed + 2 warm-up + 1 profiled forward
python tools/rocpd_summary.py $O/c3_trace/*/*_results.db > $O/${tag}_c3_kernel_stats.txt
python tools/rocpd_summary.py $O/c5_trace/*/*_results.db > $O/${tag}_c5_kernel_stats.txt
python tools/rocpd_summary.py $O/c2_trace/*/*_results.db > $O/${tag}_c2_kernel_stats.txt
python tools/rocpd_summary.py $O/c5b_trace/*/*_results.db > $O/${tag}_c5-bf16_kernel_stats.txt
python tools/stream_busy.py $O/c3_trace/*/*_results.db 6 > $O/${tag}_c3_streams.txt
# where the main stream waits inside one timed step (the last steps of a bench run are its single-stream extras: skip 6)
python tools/step_gaps.py $O/c3_trace/*/*_results.db 6 150 > $O/${tag}_c3_gaps.txt
python tools/step_gaps.py $O/c2_trace/*/*_results.db 6 150 > $O/${tag}_c2_gaps.txt
# the traffic files AND the kernel tables have to be where bench.py looks for them before the bench lines are taken (round 6:
# the roofline's kernel is the top row of the committed table of the same command)
cp $O/${tag}_c3_pmc_traffic.json $O/${tag}_c5_pmc_traffic.json $R/profiles/
cp $O/${tag}_c3_kernel_stats.txt $O/${tag}_c5_kernel_stats.txt $O/${tag}_c2_kernel_stats.txt $O/${tag}_c5-bf16_kernel_stats.txt $R/profiles/
python bench.py > $O/${tag}_bench_default.json 2> $O/bench_default.err
python bench.py --workload c5-bf16 --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $O/${tag}_bench_c5_bf16.json 2>/dev/null
python bench.py --workload c2-fwd --no-cpu-baseline --no-extra > $O/${tag}_bench_c2_fwd.json 2>/dev/null
python bench.py --workload tts --steps 20 --warmup 5 --no-extra > $O/${tag}_bench_tts.json 2>/dev/null
# the acoustic step with the adversarial term of the three spectrogram discriminators (c3-gan / c2-gan)
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/c3gan_trace -- $B --workload c3-gan --steps 3 --warmup 1 > $O/${tag}_bench_c3_gan_under_rocprof.json 2> $O/c3gan_trace.log )
python tools/rocpd_summary.py $O/c3gan_trace/*/*_results.db > $O/${tag}_c3_gan_kernel_stats.txt
rm -rf $O/c3gan_trace
python bench.py --workload c3-gan --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/${tag}_bench_c3_gan.json 2>/dev/null
python bench.py --workload c2-gan --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/${tag}_bench_c2_gan.json 2>/dev/null
python bench.py --workload c3-textual --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/${tag}_bench_c3_textual.json 2>/dev/null
python bench.py --workload c3-duration --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/${tag}_bench_c3_duration.json 2>/dev/null
# the phases of one c3 / c2 step UNTRACED (device time stamps on the streams: the tracer serialises the two encoders)
for wl in c3 c2; do
  STY_STEP_PROBE=1 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 8 --warmup 3 2>/dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], "step", round(d["ms_per_step"], 3), "ms; device time stamps of one step (ms since its first launch):")
for n, t in d.get("phases_ms", []): print(f"{t:9.3f}  {n}")
' $wl >> $O/${tag}_c3_phases.txt
done
# the c3 step serialised (style encoder on the main stream, weight-gradient streams off): every kernel alone on the chip
( cd /tmp && STY_NO_SIDE_STREAM=1 STY_NO_SE_STREAM=1 rocprofv3 --kernel-trace --stats -d $O/serial_trace -- $B --steps 6 --warmup 2 > /dev/null 2> $O/serial_trace.log )
python tools/rocpd_summary.py $O/serial_trace/*/*_results.db > $O/${tag}_c3_serial_kernel_stats.txt
rm -rf $O/serial_trace
bash tools/probes/fft_variants.sh > $O/${tag}_fft_variants.txt 2>&1
# matrix-pipe utilisation counters (SQ_VALU_MFMA_BUSY_CYCLES) of the serialised c3 step -> gpurun_out/pmc_mfma/${tag}_c3_pmc_mfma.txt
bash tools/pmc_mfma.sh $tag; cp $R/gpurun_out/pmc_mfma/${tag}_c3_pmc_mfma.txt $O/ 2>/dev/null
python tools/convp16_bench.py 10 2>/dev/null | grep conv > $O/${tag}_convp16_microbench.txt
# round 6: the twin-operand conv against convp16_kernel<.., X16> (A/B, bit-equality), its phases, the per-shape table of the serial step
python tools/convq_bench.py 10 > $O/${tag}_convq_microbench.txt 2>/dev/null
python tools/convq_phases.py 10 > $O/${tag}_convq_phases.txt 2>/dev/null
bash tools/shapes_c3.sh $tag > /dev/null 2>&1; cp $R/gpurun_out/shapes_$tag.txt $O/${tag}_c3_shapes.txt 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/overlap_probe.hip -o /tmp/overlap_probe 2>/dev/null && /tmp/overlap_probe > $O/${tag}_overlap_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/buffer_offset_probe.hip -o /tmp/buffer_offset_probe 2>/dev/null && /tmp/buffer_offset_probe > $O/${tag}_buffer_offset_probe.txt
rm -rf $O/c3_trace $O/c5_trace $O/c2_trace $O/c5b_trace $O/c3_fetch/*/*agent_info.csv $O/c5_fetch/*/*agent_info.csv
ls -la $O
