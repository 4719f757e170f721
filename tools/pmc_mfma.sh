#!/bin/bash
# matrix-core utilisation counters of the c3 step, every kernel alone on the chip (side streams off); one --pmc pass
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mfma
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=2 STY_NO_SIDE_STREAM=1 STY_NO_SE_STREAM=1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/pass -- python $R/bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 1 > /dev/null 2> $O/pass.log
cd $R
python tools/pmc_mfma.py $O/pass > $O/${tag}_c3_pmc_mfma.txt 2> $O/summary.err
rm -rf $O/pass
