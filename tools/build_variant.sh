#!/bin/bash
# A second copy of the library for same-session A/B timing: the named sources recompiled from the working tree with extra hipcc
# flags (-D switches of experimental variants), everything else from the current build.
#   bash tools/build_variant.sh <name> "<file1.hip file2.hip ...>" [extra hipcc flags ...]   ->  tools/bin/libddsp_amd_<name>.so
#   python tools/with_lib.py tools/bin/libddsp_amd_<name>.so <script.py> [args]               runs a script against it
set -e
cd "$(dirname "$0")/.."
NAME=$1; FILES=$2; shift; shift
python -c "from ddsp_amd import build; build.build(verbose=False)"
mkdir -p tools/bin
objs=""
for o in ddsp_amd/lib/obj/*.o; do
  b=$(basename $o .o)
  if echo " $FILES " | grep -q " $b.hip "; then
    extra=$(python -c "from ddsp_amd import build; print(' '.join(build.EXTRA_FLAGS.get('$b.hip', [])))")      # the product's per-source switches
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude $extra "$@" -c ddsp_amd/csrc/$b.hip -o tools/bin/${b}_$NAME.o &
    objs="$objs tools/bin/${b}_$NAME.o"
  else
    objs="$objs $o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/bin/libddsp_amd_$NAME.so
echo "built tools/bin/libddsp_amd_$NAME.so ($FILES $@)"
