set -x
cd /root/repo
python bench.py > gpurun_out/bench_v6.json 2> gpurun_out/bench_v6.err
cd /tmp && export TMPDIR=/tmp
W="--extra none --no-cpu-baseline --steps 30 --warmup 8"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o k -- python /root/repo/bench.py $W > /tmp/ks.out 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/kernel_stats_v6_fp16.csv
python /root/repo/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > /root/repo/gpurun_out/step_timeline_v6_fp16.txt
grep "^{" /tmp/ks.out | tail -1 > /root/repo/gpurun_out/bench_under_rocprof_v6.json
P="--extra none --no-cpu-baseline --steps 12 --warmup 8"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python /root/repo/bench.py $P > /tmp/pf.out 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o p -- python /root/repo/bench.py $P > /tmp/pw.out 2>&1
python /root/repo/tools/pmc_all.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) /root/repo/gpurun_out/pmc_v6_fp16.json --precision fp16 --workload "python bench.py $P"
