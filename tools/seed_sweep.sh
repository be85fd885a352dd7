#!/bin/bash
# The `-m gpu` suite under several HCTR_TEST_SEEDs on one box (tests/conftest.py seeds every test from
# its node id xor this value).   bash tools/seed_sweep.sh TAG "0 1 2 3 4"   -> gpurun_out/TAG_seed*.txt
TAG=${1:-r6_gpu_suite}
SEEDS=${2:-"0 1 2 3 4"}
mkdir -p gpurun_out
rc=0
for s in $SEEDS; do
  HCTR_TEST_SEED=$s timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/${TAG}_seed$s.txt 2>&1 || rc=1
  echo "seed $s: $(tail -1 gpurun_out/${TAG}_seed$s.txt)"
done
exit $rc
