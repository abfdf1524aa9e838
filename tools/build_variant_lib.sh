#!/bin/bash
# A second copy of the library for same-session A/B timing (tools/exp_ab.py): the two default kernels' (and the loss's) sources taken from
# a git revision (default: the working tree), everything else from the current build.  Boxes of the pool differ by up
# to 6 %, so two versions of a kernel are only comparable inside one gpurun call.
#   bash tools/build_variant_lib.sh <name> [git-rev] [extra hipcc flags ...]   ->  tools/bin/libddsp_amd_<name>.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; REV=${2:-WORKTREE}; shift; shift || true
python -c "from ddsp_amd import build; build.build(verbose=False)"
mkdir -p tools/bin
SRC=/tmp/ddsp_variant_$NAME
rm -rf $SRC; mkdir -p $SRC/ddsp_amd/csrc $SRC/include
if [ "$REV" = "WORKTREE" ]; then
  cp ddsp_amd/csrc/* $SRC/ddsp_amd/csrc/; cp include/* $SRC/include/
else
  git archive $REV ddsp_amd/csrc include | tar -x -C $SRC
fi
for f in harmonic harmonic_table harmonic_bwd_table filtered_noise_mfma filtered_noise_general spectral_loss; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I$SRC/include "$@" -c $SRC/ddsp_amd/csrc/$f.hip -o tools/bin/${f}_$NAME.o &
done
wait
objs=$(ls ddsp_amd/lib/obj/*.o | grep -v "/harmonic.o\|harmonic_table.o\|harmonic_bwd_table.o\|filtered_noise_mfma.o\|filtered_noise_general.o\|spectral_loss.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/bin/harmonic_$NAME.o tools/bin/harmonic_table_$NAME.o tools/bin/harmonic_bwd_table_$NAME.o tools/bin/filtered_noise_mfma_$NAME.o tools/bin/filtered_noise_general_$NAME.o tools/bin/spectral_loss_$NAME.o -o tools/bin/libddsp_amd_$NAME.so
echo "built tools/bin/libddsp_amd_$NAME.so from $REV"
