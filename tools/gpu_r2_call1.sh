#!/bin/bash
# Round 2, GPU call 1: the never-run tests first (no -x), measured parity errors, the prepared variants, microbench, benches.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export DDSP_PARITY_LOG=$PWD/$OUT/parity_errors.jsonl
rm -f $DDSP_PARITY_LOG
echo "== lscpu / mem"; nproc; grep -m1 MemAvailable /proc/meminfo; lscpu | grep -E "Thread|Core|Socket|Model name" | head
echo "== microbench6 (unaligned LDS fragment reads)"
timeout 60 tools/microbench6 2>&1 | tee $OUT/microbench_lds_unaligned.txt
echo "== pytest general (never run on hardware before), no -x"
timeout 900 python -m pytest tests/test_gpu_parity_general.py -m gpu -q 2>&1 | tail -40 | tee $OUT/pytest_general.txt
echo "== pytest parity, no -x"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -40 | tee $OUT/pytest_parity.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
echo "== bench default (as the driver runs it: --steps 20 --warmup 5)"
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench_driver.err | tail -1 | tee $OUT/bench_driver_like.json | cut -c1-600
echo "== bench default (1000 steps)"
timeout 600 python bench.py --no-cpu-baseline --also-other-mode 2>$OUT/bench_1000.err | tail -1 | tee $OUT/bench_1000.json | cut -c1-400
echo "== Harmonic: wavetable kernel vs direct sum (+ tphase variant)"
timeout 120 python tools/exp_table.py 32 128 2>&1 | tail -2 | tee $OUT/harm_table_vs_direct.json
echo "== FilteredNoise IR design: vector vs matrix vs matrix_direct"
timeout 120 python tools/exp_noise_ir.py 32 128 2>&1 | tail -2 | tee $OUT/noise_ir_variants.json
echo "== timelines"
timeout 120 python tools/exp_table_timeline.py 32 2>&1 | grep -A40 "launch 2" | tee $OUT/timeline_harm_table_b32.txt | head -12
DDSP_EXP_TABLE_PHASE_ON_T=1 timeout 120 python tools/exp_table_timeline.py 32 2>&1 | grep -A40 "launch 2" | tee $OUT/timeline_harm_table_tphase_b32.txt | head -3
for FL in 0x0 0x2 0x6; do for B in 32 128; do timeout 120 python tools/exp_timeline_noise.py $B $FL 2>&1 | tail -9 | tee -a $OUT/timeline_noise_variants.txt | tail -4; done; done
echo "== streaming latency"
timeout 300 python tools/bench_streaming.py 2>&1 | tail -1 | tee $OUT/bench_streaming.json
echo "== done"
