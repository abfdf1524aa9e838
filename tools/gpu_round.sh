#!/bin/bash
# One GPU-box session for the round's evidence: parity tests, smoke, bench, rocprof kernel traces, PMC traffic, SQ counters.
# Usage (from the build container): gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export DDSP_PARITY_LOG=$PWD/$OUT/parity_errors.jsonl
rm -f $DDSP_PARITY_LOG
echo "== pytest -m gpu (general first, no -x)"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tee $OUT/pytest_gpu_full.txt | tail -8
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
echo "== bench, as the driver runs it (--steps 20 --warmup 5)"
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench_driver.err | tail -1 > $OUT/bench_driver_like.json; cut -c1-300 $OUT/bench_driver_like.json
echo "== bench (defaults: 1000 steps)"
timeout 600 python bench.py --no-cpu-baseline --also-other-mode 2>$OUT/bench_1000.err | tail -1 > $OUT/bench_1000.json
python - <<PY
import json
d = json.load(open('$OUT/bench_1000.json'))
print('B=%d:' % d['config']['batch_per_gpu'], round(d['value']), 'Msamples/s', round(d['ms_per_step'] * 1e3, 2), 'us  whole-step frac', round(d['roofline']['whole_step']['frac'], 4),
      'dominant', d['roofline']['kernel'], round(d['roofline']['avg_launch_us'], 2), 'us frac', round(d['roofline']['frac'], 4), 'one stream', d.get('one_stream'), 'other mode', d.get('other_issue_mode'))
print('isolated', d['kernel_breakdown_us_isolated'])
print('fused add', d.get('fused_add'))
c1 = d['configs_1']
print('configs[1] B=32:', round(c1['value']), round(c1['ms_per_step'] * 1e3, 2), 'us frac', round(c1['whole_step']['frac'], 4), c1['kernel_breakdown_us'], c1.get('dominant_kernel', {}).get('frac'))
c4 = d.get('configs_4')
if c4: print('configs[4] B=32 48 kHz K=200:', round(c4['value']), round(c4['ms_per_step'] * 1e3, 2), 'us frac', round(c4['whole_step']['frac'], 4), c4['kernel_breakdown_us'], 'one stream', c4.get('one_stream'))
PY
echo "== rocprofv3 kernel trace (configs[1], batch 32, two streams)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --batch 32 --no-cpu-baseline --no-aux --no-second-shape > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1 )
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats_b32.csv; head -6 $f | cut -c1-200; done
grep "^{\"metric\"" $OUT/rocprof_bench.log | tail -1 > $OUT/bench_b32_under_rocprof.json
echo "== rocprofv3 kernel trace, batch 128 (north-star shape, one stream)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof128 -o trace -- python $GRAFT_REPO_ROOT/bench.py --batch 128 --streams 1 --no-cpu-baseline --no-aux --no-second-shape > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_b128.log 2>&1 )
for f in $(find $OUT/prof128 -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats_b128.csv; head -6 $f | cut -c1-200; done
grep "^{\"metric\"" $OUT/rocprof_bench_b128.log | tail -1 > $OUT/bench_b128_under_rocprof.json
rm -rf $OUT/prof $OUT/prof128
echo "== configs[4] per GPU (48 kHz, 200 harmonics, 10 s, batch 32): rocprofv3 kernel trace, one stream; then BASELINE's other configurations"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof5 -o trace -- python $GRAFT_REPO_ROOT/bench.py --batch 32 --n-frames 2500 --n-harmonics 200 --n-samples 480000 --sample-rate 48000 --steps 200 --warmup 60 --streams 1 --no-cpu-baseline --no-aux --no-second-shape > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_config5.log 2>&1 )
for f in $(find $OUT/prof5 -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats_config5.csv; head -4 $f | cut -c1-200; done
grep "^{\"metric\"" $OUT/rocprof_bench_config5.log | tail -1 > $OUT/bench_config5_under_rocprof.json
rm -rf $OUT/prof5
# (BASELINE's other configurations are blocks of the bench line itself since round 5: configs_2, configs_3)
python - <<PY
import json
d = json.load(open('$OUT/bench_1000.json'))
for k in ('configs_2', 'configs_3', 'fnoise_11_bit_levels', 'fnoise_full_resolution'):
  b = d.get(k, {})
  print(k, {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in b.items() if kk in ('ms_per_step', 'value', 'error', 'cost_of_the_twelve_bits_us', 'with_gradient_wrt_audio', 'kernel_breakdown_us')}, 'frac', b.get('whole_step', {}).get('frac'))
PY
echo "== PMC: HBM traffic per launch (separate FETCH_SIZE / WRITE_SIZE passes), batch 32 and 128"
bash tools/pmc_traffic.sh $TAG/pmc_traffic_b32 32 > /dev/null 2>&1; cp $OUT/pmc_traffic_b32/pmc_traffic.json $OUT/pmc_traffic.json; python -c "import json; d=json.load(open('$OUT/pmc_traffic.json')); print(d['kernels'])"
bash tools/pmc_traffic.sh $TAG/pmc_traffic_b128 128 > /dev/null 2>&1; cp $OUT/pmc_traffic_b128/pmc_traffic.json $OUT/pmc_traffic_b128.json; python -c "import json; d=json.load(open('$OUT/pmc_traffic_b128.json')); print(d['kernels'])"
echo "== PMC: HBM traffic per launch, configs[4] per GPU (48 kHz, 200 harmonics, 2500 frames of 192, batch 32)"
bash tools/pmc_traffic.sh $TAG/pmc_traffic_config5 32 2500 200 480000 48000 > /dev/null 2>&1; cp $OUT/pmc_traffic_config5/pmc_traffic.json $OUT/pmc_traffic_config5.json; python -c "import json; d=json.load(open('$OUT/pmc_traffic_config5.json')); print(d.get('shape'), d['kernels'])"
echo "== PMC: SQ counters, batch 128"
bash tools/pmc.sh $TAG/pmc_sq_b128 128 > $OUT/pmc_sq_counters_b128.log 2>&1; cp $OUT/pmc_sq_b128/summary.txt $OUT/pmc_sq_counters_b128.txt 2>/dev/null; python tools/pmc_summary.py $OUT/pmc_sq_b128 --json $OUT/pmc_sq_b128.json 128 1000 100 64000 16000 > /dev/null 2>&1; grep -A30 "== harm_table\|== noise_mfma" $OUT/pmc_sq_counters_b128.txt | grep "==\|SQ_WAIT_ANY\|SQ_WAVE_CYCLES\|SQ_LDS_BANK\|SQ_LDS_IDX\|SQ_INSTS_VALU \|SQ_ACTIVE_INST_ANY\|SQ_WAIT_INST_ANY" | head -20
rm -rf $OUT/pmc_traffic_b32 $OUT/pmc_traffic_b128 $OUT/pmc_traffic_config5 $OUT/pmc_sq_b128
echo "== next-row benches (Reverb, SpectralLoss, backward, streaming)"
timeout 300 python tools/bench_reverb.py 32 2>&1 | tail -1 | tee $OUT/bench_reverb_b32.json | cut -c1-200
timeout 300 python tools/bench_reverb.py 128 64000 48000 1 2>&1 | tail -1 | tee $OUT/bench_reverb_b128_one_ir.json | cut -c1-260
timeout 300 python tools/bench_reverb.py 128 2>&1 | tail -1 | tee $OUT/bench_reverb_b128_ir_per_row.json | cut -c1-260
timeout 300 python tools/bench_spectral_loss.py 128 2>&1 | tail -1 | tee $OUT/bench_spectral_loss_b128.json | cut -c1-200
timeout 300 python tools/bench_spectral_loss.py 32 2>&1 | tail -1 | tee $OUT/bench_spectral_loss_b32.json | cut -c1-200
timeout 300 python tools/bench_backward.py 32 2>&1 | tail -1 | tee $OUT/bench_backward_b32.json | cut -c1-300
timeout 300 python tools/bench_streaming.py 2>&1 | tail -1 | tee $OUT/bench_streaming.json | cut -c1-300
echo "== rocprofv3 kernel trace of the Reverb and SpectralLoss benches at batch 128 (BASELINE configs[3] / configs[2]'s own kernels)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/profrv -o trace -- python $GRAFT_REPO_ROOT/tools/bench_reverb.py 128 64000 48000 1 > /dev/null 2>&1 )
for f in $(find $OUT/profrv -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats_reverb_b128_one_ir.csv; head -5 $f | cut -c1-160; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/profsl -o trace -- python $GRAFT_REPO_ROOT/tools/bench_spectral_loss.py 128 > /dev/null 2>&1 )
for f in $(find $OUT/profsl -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats_spectral_loss_b128.csv; head -5 $f | cut -c1-160; done
rm -rf $OUT/profrv $OUT/profsl
echo "== shapes behind the specialised paths (tools/bench_generic.py), batch 32 and 128"
timeout 300 python tools/bench_generic.py 32 2>/dev/null | grep "^{" > $OUT/generic_shapes_b32.jsonl; cut -c1-200 $OUT/generic_shapes_b32.jsonl
echo "== determinism stress: ${STRESS_ITERS:-300} launches per case, bits compared on the device (tools/stress_determinism.py)"
timeout 600 python tools/stress_determinism.py --iters ${STRESS_ITERS:-300} --label $TAG --out $OUT/determinism_stress.jsonl 2>&1 | grep "SUMMARY\|MISMATCH" | cut -c1-300
echo "== done"
