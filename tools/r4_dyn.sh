cd /root/repo
for CFG in "HCTR_HT_FINISH_BLOCKS=256" "HCTR_HT_FINISH_BLOCKS=128"; do
echo "== $CFG"
env $CFG python bench.py --extra next --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=j['extra']['ebc_dynamic_multi_hot']; print(d.get('forward_us'), d.get('backward_update_us'), d.get('forward_backward_update_us'), d.get('error'))"
done
