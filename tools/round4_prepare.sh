#!/bin/bash
# (no GPU) before the first gpurun call of the next round: cross-compile the prepared kernel variants
# next to the product library (they travel to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")/.."
python -c 'import __graft_entry__ as g; g.build()'
for v in index_segments index_segments_v2 sort_first_pass; do
  make -C hugectr_amd/csrc -j 8 VARIANT=$PWD/tools/wip/$v TAG=$v > /tmp/build_$v.log 2>&1 || { tail -20 /tmp/build_$v.log; exit 1; }
  ls -la hugectr_amd/libhugectr_amd_$v.so
done
echo "next: gpurun --timeout 2400 -- 'bash tools/round4_first_call.sh'"
