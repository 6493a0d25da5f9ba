# binary search for the first device allocation whose poisoning breaks the given pytest selection (VS_TEST_POISON byte in $1)
B=${1:-0x7F}; shift
run() { VS_TEST_POISON=$B VS_POISON_UPTO=$1 timeout -k 5 120 python -m pytest tests/test_golden.py tests/test_gpu_api.py -m gpu -q -x -k "golden_vectors or drop_in or class_on_gpu or tracking_harness" > /tmp/ps.log 2>&1; }
lo=0; hi=200
run $hi && { echo "does not fail with the first $hi allocations poisoned"; exit 0; }
while [ $((hi - lo)) -gt 1 ]; do
  mid=$(( (lo + hi) / 2 ))
  if run $mid; then lo=$mid; echo "upto $mid: passes"; else hi=$mid; echo "upto $mid: FAILS"; fi
done
echo "the allocation whose poisoning breaks it: ordinal $lo (0-based)"
VS_TEST_POISON=$B VS_POISON_UPTO=$hi VS_POISON_LOG=1 timeout -k 5 120 python -m pytest tests/test_golden.py tests/test_gpu_api.py -m gpu -q -x -k "golden_vectors or drop_in or class_on_gpu or tracking_harness" 2>&1 | grep "\[poison\]" | sed -n "$((lo > 3 ? lo - 3 : 1)),$((lo + 2))p"
