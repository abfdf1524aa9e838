#!/bin/bash
# One GPU-box session: microbench, parity tests, smoke, bench, rocprof kernel trace.
# Usage (from the build container): gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
if [ -x tools/microbench ]; then echo "== microbench"; timeout 120 tools/microbench 2>&1 | tee $OUT/microbench.txt; fi
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench (batch 32)"
timeout 600 python bench.py --also-other-mode 2>&1 | tail -3 | tee $OUT/bench_b32.json
echo "== bench (batch 128)"
timeout 600 python bench.py --batch 128 --no-cpu-baseline --also-other-mode 2>&1 | tail -3 | tee $OUT/bench_b128.json
echo "== rocprofv3 kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-aux > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1 )
ls -R $OUT/prof | head -20
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -2); do echo "--- $f"; head -15 $f; done
echo "== rocprofv3 kernel trace, batch 128 (north-star shape: one stream, harm_table_kernel dominant)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof128 -o trace -- python $GRAFT_REPO_ROOT/bench.py --batch 128 --no-cpu-baseline --no-aux > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_b128.log 2>&1 )
for f in $(find $OUT/prof128 -name "*kernel_stats*.csv" | head -1); do echo "--- $f"; head -8 $f | cut -c1-220; done
echo "== Harmonic: wavetable kernel vs direct sum, and the wavetable kernel's per-tick timeline"
timeout 120 python tools/exp_table.py 32 128 2>&1 | tail -2 | tee $OUT/harm_table_vs_direct.json
timeout 120 python tools/exp_table_timeline.py 32 2>&1 | grep -A40 "launch 2" | tee $OUT/timeline_harm_table_b32.txt | head -5
echo "== torchrun world=1 sanity (the N>1 code path of bench.py)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300
echo "== next-row benches (Reverb, SpectralLoss, backward)"
timeout 300 python tools/bench_reverb.py 32 2>&1 | tail -1 | tee $OUT/bench_reverb_b32.json
timeout 300 python tools/bench_reverb.py 128 2>&1 | tail -1 | tee $OUT/bench_reverb_b128.json
timeout 300 python tools/bench_spectral_loss.py 32 2>&1 | tail -1 | tee $OUT/bench_spectral_loss_b32.json
timeout 300 python tools/bench_backward.py 32 2>&1 | tail -1 | tee $OUT/bench_backward_b32.json
timeout 300 python tools/bench_backward.py 128 2>&1 | tail -1 | tee $OUT/bench_backward_b128.json
echo "== rocprofv3 kernel trace of the training step (synths + loss, forward + backward)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_train -o trace -- python $GRAFT_REPO_ROOT/tools/bench_backward.py 32 > $GRAFT_REPO_ROOT/$OUT/rocprof_train.log 2>&1 )
for f in $(find $OUT/prof_train -name "*kernel_stats*.csv" | head -1); do echo "--- $f"; head -20 $f | cut -c1-200; done
echo "== streaming latency (VST frame call)"
timeout 300 python tools/bench_streaming.py 2>&1 | tail -1 | tee $OUT/bench_streaming.json
echo "== FilteredNoise IR design: vector ALUs vs matrix cores (experimental variant)"
timeout 120 python tools/exp_noise_ir.py 32 128 2>&1 | tail -2 | tee $OUT/noise_ir_vector_vs_matrix.json
echo "== bench with the experimental kernel variants (whole step, batch 32 and 128)"
for V in "--harm-kernel table_tphase" "--noise-ir matrix" "--noise-ir matrix_direct" "--harm-kernel table_tphase --noise-ir matrix_direct"; do
  for B in 32 128; do
    timeout 300 python bench.py --batch $B --no-cpu-baseline --no-aux $V 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', 'batch', $B, 'ms_per_step', round(r['ms_per_step'],5), 'value', round(r['value']), r['kernel_breakdown_us_isolated'])" | tee -a $OUT/bench_variants.txt
  done
done
echo "== timelines of the experimental variants"
DDSP_EXP_TABLE_PHASE_ON_T=1 timeout 120 python tools/exp_table_timeline.py 32 2>&1 | grep -A40 "launch 2" | tee $OUT/timeline_harm_table_tphase_b32.txt | head -5
for FL in 0x2 0x6; do for B in 32 128; do timeout 120 python tools/exp_timeline_noise.py $B $FL 2>&1 | tail -9 | tee -a $OUT/timeline_noise_variants.txt; done; done
