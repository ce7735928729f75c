#!/bin/bash
# Re-roll every seeded GPU parity test with other seeds (tests/datagen.py: FBK_TEST_SEED): the expectations are
# computed by the oracle at run time, so every seed is a new differential test of the HIP path.
#   scripts/fuzz_parity.sh <out dir> <seed> [<seed> ...]
out=$1; shift
mkdir -p "$out"
for s in "$@"; do
  echo "== seed $s"
  FBK_TEST_SEED=$s timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider \
    --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_group.py 2>&1 | tail -15
done | tee "$out/fuzz_parity.log"
