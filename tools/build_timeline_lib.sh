#!/bin/bash
# The product library with harm_table_kernel's timeline stamps compiled in (tools/exp_table_timeline.py).
set -e
cd "$(dirname "$0")/.."
python -c "from ddsp_amd import build; build.build()"
mkdir -p tools/bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -DDDSP_WT_TIMELINE -c ddsp_amd/csrc/harmonic_table.hip -o tools/bin/harmonic_table_timeline.o
objs=$(ls ddsp_amd/lib/obj/*.o | grep -v harmonic_table.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/bin/harmonic_table_timeline.o -o tools/bin/libddsp_amd_timeline.so
