#!/bin/bash
# PMC passes (separate runs, no tracing flags besides --kernel-trace) over a short bench run.
# Usage: gpurun --timeout 900 -- 'bash tools/pmc.sh <tag> [batch]'
# PMC_CMD (optional): the command to profile instead of the short bench run, e.g. "python $GRAFT_REPO_ROOT/tools/bench_spectral_loss.py 128"
TAG=${1:-pmc}; BATCH=${2:-32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD=${PMC_CMD:-"python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-aux --no-second-shape --streams 1 --batch $BATCH"}
i=0
for SET in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY" \
  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
  "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VALU_TRANS" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pass$i -o pmc -- $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$? : $SET"
done
find $OUT -name "*counter_collection.csv" | head
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
