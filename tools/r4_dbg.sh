#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in "std_sort_order_equals" "std_sort_order_above" "sequential_sums" "rotational_histogram_equals_oracle" "with_a_floor" "switches" "two_halves or limits or reproduces"; do
  echo "== $k"
  timeout 90 python -m pytest tests/test_gpu_parity.py -q -x --timeout 80 -k "$k" 2>&1 | tail -4
done
