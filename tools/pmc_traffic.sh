#!/bin/bash
# HBM traffic per kernel launch from the TCC counters (two separate --pmc passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Usage: gpurun --timeout 900 -- 'bash tools/pmc_traffic.sh <tag> [batch [n_frames n_harmonics n_samples sample_rate]]'
TAG=${1:-pmc_traffic}; BATCH=${2:-32}
NF=${3:-1000}; NH=${4:-100}; NS=${5:-64000}; SR=${6:-16000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-aux --no-second-shape --streams 1 --batch $BATCH --n-frames $NF --n-harmonics $NH --n-samples $NS --sample-rate $SR"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o pmc -- $CMD > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT $BATCH $NF $NH $NS $SR | tee $OUT/pmc_traffic.json
