#!/bin/bash
# round 5: the bench line, a kernel trace of the same command, the PMC traffic passes and the SQ
# pass, on ONE box.   bash tools/measure_round.sh TAG [all]   (files land in gpurun_out/r6_*_TAG.*;
# copy what is to be judged into profiles/).  Counter files are stamped with the hash of the kernel
# sources (tools/csrc_hash.py) -- bench.py quotes them only while that hash still matches.
set -x
TAG=${1:-v1}
cd /root/repo
COMMIT=$(git rev-parse --short HEAD 2>/dev/null || cat tools/.commit 2>/dev/null || echo unknown)
HASH=$(python tools/csrc_hash.py)
cd /tmp && export TMPDIR=/tmp
W="--extra none --no-cpu-baseline --steps 30 --warmup 8"
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o k -- python /root/repo/bench.py $W > /tmp/ks.out 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r6_bench_n1_kernel_stats_$TAG.csv
python /root/repo/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > /root/repo/gpurun_out/r6_step_timeline_$TAG.txt
grep "^{" /tmp/ks.out | tail -1 > /root/repo/gpurun_out/r6_bench_under_rocprof_$TAG.json
pmc() {  # label, alpha, bench flags
  L=$1; A=$2; shift 2
  rm -rf /tmp/pf /tmp/pw
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python /root/repo/bench.py "$@" > /tmp/pf.out 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o p -- python /root/repo/bench.py "$@" > /tmp/pw.out 2>&1
  python /root/repo/tools/pmc_all.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) /root/repo/gpurun_out/r6_pmc_hbm_traffic_$L.json --precision fp16 --alpha $A --commit $COMMIT --csrc-hash $HASH --workload "python bench.py $*"
}
P="--no-cpu-baseline --steps 12 --warmup 8"
pmc fp16 1.1 --extra none $P
# SQ pass (matrix-pipe utilisation, wait split) -- its own run, kernel trace only
rm -rf /tmp/psq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/psq -o p -- python /root/repo/bench.py --extra none $P > /tmp/psq.out 2>&1
python /root/repo/tools/pmc_sq.py $(find /tmp/psq -name "*counter_collection.csv" | head -1) /root/repo/gpurun_out/r6_pmc_sq_counters_$TAG.json --commit $COMMIT --csrc-hash $HASH --workload "python bench.py --extra none $P"
if [ "$2" = "all" ]; then
  pmc uniform_big_tables 0.0 --extra uniform --steps 2 --warmup 2 --extra-steps 10 --no-cpu-baseline
  pmc ebc 1.1 --extra ebc --steps 2 --warmup 2 --extra-steps 6 --no-cpu-baseline
fi
# the bench line last: it quotes the counter files of THIS run (same sources, same box)
cd /root/repo
cp gpurun_out/r6_pmc_hbm_traffic_fp16.json profiles/ 2>/dev/null
cp gpurun_out/r6_pmc_hbm_traffic_uniform_big_tables.json gpurun_out/r6_pmc_hbm_traffic_ebc.json profiles/ 2>/dev/null
cp gpurun_out/r6_pmc_sq_counters_$TAG.json profiles/r6_pmc_sq_counters.json
python bench.py --extra-file gpurun_out/r6_bench_extra_$TAG.json > gpurun_out/r6_bench_n1_$TAG.json 2> gpurun_out/r6_bench_n1_$TAG.err
