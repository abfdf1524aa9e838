#!/bin/bash
# Round 2, GPU call 2: the matrix-core FilteredNoise kernel - parity, A/B timing against the vector-ALU FIR, timeline, bench.
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export DDSP_PARITY_LOG=$PWD/$OUT/parity_errors.jsonl
rm -f $DDSP_PARITY_LOG
echo "== pytest (noise / DAG / smoke-relevant cases), no -x"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_general.py -m gpu -q -k "noise or processor_group or full_size or fir_filter or vst_dag or training_loop" 2>&1 | tee $OUT/pytest_noise_full.txt | tail -15
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
echo "== FilteredNoise: matrix-core kernel vs vector-ALU FIR"
timeout 300 python tools/exp_noise_fir.py 32 128 2>&1 | tail -40 | tee $OUT/noise_mfma_vs_vector.txt
echo "== Harmonic (chunk-level Nyquist flag)"
timeout 120 python tools/exp_table.py 32 128 2>&1 | tail -2 | tee $OUT/harm_table_vs_direct.json
echo "== bench (1000 steps), then driver-like"
timeout 600 python bench.py --no-cpu-baseline --also-other-mode 2>$OUT/bench_1000.err | tail -1 | tee $OUT/bench_1000.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('B=32:', round(d['value']), 'Msamples/s', d['ms_per_step'], 'ms  whole-step frac', round(d['roofline']['whole_step']['frac'], 4), 'dominant', d['roofline']['kernel'], round(d['roofline']['avg_launch_us'], 2), 'us', 'other mode', d.get('other_issue_mode'))
print('iso', d['kernel_breakdown_us_isolated'])
ns = d['north_star_shape']; print('B=128:', round(ns['value']), ns['ms_per_step'], 'frac', round(ns['whole_step']['frac'], 4), ns['kernel_breakdown_us'])
"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/bench_driver.err | tail -1 | tee $OUT/bench_driver_like.json | cut -c1-200
echo "== done"
