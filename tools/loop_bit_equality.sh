#!/bin/bash
# The two row-alone-equals-row-in-its-batch tests, N times (default 20): the ones that failed once, on one box, with the second
# attempt at the Nyquist-crossing path (profiles/r03w_f0_dependence_two_effects.txt) - loop them while bisecting that change.
#   gpurun --timeout 600 -- 'bash tools/loop_bit_equality.sh 20'
N=${1:-20}
fail=0
for i in $(seq 1 $N); do
  out=$(timeout 120 python -m pytest tests/test_gpu_contract_shapes.py::test_north_star_shape_batch128_harmonic \
        "tests/test_gpu_parity.py::test_full_size_properties_batch32" -q 2>&1 | tail -1)
  echo "$i: $out"
  case "$out" in *failed*) fail=$((fail + 1));; esac
done
echo "runs with a failure: $fail of $N"
