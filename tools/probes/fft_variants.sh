#!/bin/bash
# kernel times of the FFT front end at c3 size (B = 32, 6.5 s) for the tile / XCD-mapping variants (rocprofv3 --kernel-trace)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "tf16" "tf8:STY_FFT_TF=8" "tf32:STY_FFT_TF=32" "tf16x:STY_FFT_XCD=1" "tf32x:STY_FFT_TF=32 STY_FFT_XCD=1" "tf8x:STY_FFT_TF=8 STY_FFT_XCD=1"; do
  name=${v%%:*}; envs=""; [[ "$v" == *:* ]] && envs=${v#*:}
  rm -rf /tmp/fftv_$name
  env $envs PYTHONPATH=$R rocprofv3 --kernel-trace --stats -d /tmp/fftv_$name -- python $R/tests/frontend_ab_worker.py /tmp/fftv_$name.pt 32 156000 > /dev/null 2> /tmp/fftv_$name.log
  echo "== $name ($envs)"
  python $R/tools/rocpd_summary.py /tmp/fftv_$name/*/*_results.db 2>/dev/null | grep -E "fft|magphase|frame_bwd|fb_sparse" | cut -c1-130
done
