#!/bin/bash
# Tuning aid: build libstylish_hip_<name>.so with extra -D flags for selected sources, next to the product library.
#   tools/build_variant.sh pipe0 "-DSTY_PIPE=0" conv1d.hip
# Select at run time with STY_LIB_VARIANT=<name> (stylish_tts_amd/lib.py).
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/stylish_tts_amd
python -m stylish_tts_amd.build >/dev/null
objs=()
for o in $pkg/build/*.o; do
  base=$(basename $o .o)
  use=$o
  for s in "$@"; do
    if [ "$s" == "$base.hip" ]; then
      use=$pkg/build/$base.$name.vo
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c $pkg/csrc/$s -o $use
    fi
  done
  objs+=($use)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/libstylish_hip_$name.so "${objs[@]}"
echo $pkg/libstylish_hip_$name.so
