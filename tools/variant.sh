#!/bin/bash
# A kernel variant = a directory of files laid over hugectr_amd/csrc (tools/wip/<name>/).
#   tools/variant.sh check <dir>   (no GPU) the CPU pre-flight: the kernels' source of the variant
#                                  under the host interpreter (tests/emu) -- test_emu_cpu.py and the
#                                  index / embedding / update `-m gpu` tests with HCTR_EMU=1 --, then
#                                  the cross-compile of libhugectr_amd_<name>.so for gfx950
#   tools/variant.sh gpu <name>    (on the GPU box, inside gpurun) the same `-m gpu` tests on the
#                                  hardware with HCTR_LIB_VARIANT=<name>, then the kernel medians of
#                                  the bench's main leg for the product and for the variant
#                                  (gpurun_out/variant_<name>.txt)
set -e
cd "$(dirname "$0")/.."
FILES="tests/test_hash_gpu.py tests/test_embedding_gpu.py tests/test_golden_gpu.py tests/test_sort_gpu.py tests/test_det_gpu.py tests/test_ebc_dynamic_gpu.py"
case "$1" in
check)
  D=$(realpath "$2"); N=$(basename "$D")
  HCTR_EMU_VARIANT=$D python -m pytest tests/test_emu_cpu.py tests/test_ref_gpu_kernels_cpu.py tests/test_ref_hashtable_cpu.py tests/test_ref_static_table_cpu.py -x -q
  HCTR_EMU_VARIANT=$D python tests/emu/fuzz_ref_kernels.py --seed 7 --cases 150
  HCTR_EMU=1 HCTR_EMU_VARIANT=$D python -m pytest $FILES -x -q -m gpu -n 4 --timeout 900 -p no:cacheprovider
  make -C hugectr_amd/csrc -j 8 VARIANT=$D TAG=$N > /tmp/variant_build.log 2>&1 || { tail -30 /tmp/variant_build.log; exit 1; }
  echo "built hugectr_amd/libhugectr_amd_$N.so; next: gpurun --timeout 900 -- 'bash tools/variant.sh gpu $N'"
  ;;
gpu)
  N=$2; mkdir -p gpurun_out; OUT=gpurun_out/variant_$N.txt; : > $OUT
  HCTR_LIB_VARIANT=$N timeout 700 python -m pytest $FILES -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT
  for V in "" "$N"; do
    echo "== library: ${V:-product}" >> $OUT
    HCTR_LIB_VARIANT=$V timeout 300 bash tools/profile_index_stage.sh 2>&1 | tail -8 >> $OUT
    grep "^{" /tmp/ks.out | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'], 'index', j.get('roofline_index',{}).get('us'), 'update', j.get('roofline_update',{}).get('us'))" >> $OUT 2>&1 || true
  done
  cat $OUT
  ;;
*) echo "usage: $0 check <dir> | gpu <name>"; exit 2;;
esac
