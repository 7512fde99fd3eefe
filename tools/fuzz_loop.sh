#!/bin/bash
# tools/fuzz_loop.sh <first> <last> -- run the seeded GPU fuzz tests for a range of S2P_FUZZ_SEED values.
cd "$(dirname "$0")/.."
fail=0
for s in $(seq $1 $2); do
  S2P_FUZZ_SEED=$s python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1 | grep -q passed || { echo "seed $s FAILED"; fail=1; }
done
echo "fuzz seeds $1..$2 done, fail=$fail"
