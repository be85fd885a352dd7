#!/bin/bash
# (on the GPU box) round 4 measurement call: the update / index / sort tests, the main bench leg
# with the hot-row path on and off, and a kernel trace + step timeline of the main leg.
#   bash tools/r4_call.sh TAG [full]     -> gpurun_out/r4_*_TAG.*
TAG=${1:-a}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4_summary_$TAG.txt; : > $OUT
if [ "$2" = "full" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee -a $OUT
else
  timeout 600 python -m pytest tests/test_embedding_gpu.py tests/test_sort_gpu.py tests/test_hash_gpu.py tests/test_golden_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee -a $OUT
fi
W="--extra none --no-cpu-baseline --steps 30 --warmup 8"
line() { python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],4), 'update', j['roofline_update']['us'], j['roofline_update']['frac'], 'index', j['roofline_index']['us'], 'stages', j['stage_us_per_step'], 'steady', j['stage_us_per_step_no_new_keys'], 'loss', j['config']['final_loss'])"; }
for H in 8192 0 ${R4_EXTRA_H}; do
  echo "== HCTR_HOT_ROWS=$H" >> $OUT
  HCTR_HOT_ROWS=$H timeout 300 python bench.py $W 2>gpurun_out/r4_bench_err_$TAG.txt | grep "^{" | tail -1 | line >> $OUT 2>&1
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o k -- python /root/repo/bench.py $W > /tmp/ks.out 2>&1
cd /root/repo
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) gpurun_out/r4_bench_n1_kernel_stats_$TAG.csv
python tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > gpurun_out/r4_step_timeline_$TAG.txt 2>&1
grep "^{" /tmp/ks.out | tail -1 > gpurun_out/r4_bench_under_rocprof_$TAG.json
echo "== timeline (update part)" >> $OUT
grep -n "expand_pairs\|rs_\|hot_\|seg_\|ht_" gpurun_out/r4_step_timeline_$TAG.txt >> $OUT
echo "== sort microbench" >> $OUT; timeout 120 python tools/microbench_sort.py >> $OUT 2>&1; cat $OUT
