cd /root/repo
cat > /tmp/tier_drv.py <<'P'
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
r = bench.tiered_leg(6, 3, torch.device("cuda", 0))
print(r["lookup_us"], r["update_us"])
P
for PC in 1 4; do
  rm -rf /tmp/kt$PC
  ( cd /tmp && export TMPDIR=/tmp && HCTR_TIER_PIECES=$PC rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$PC -o k -- python /tmp/tier_drv.py > /tmp/kt$PC.out 2>&1 )
  echo "==== pieces $PC: $(tail -1 /tmp/kt$PC.out)"
  python - <<P
import csv,glob
f=glob.glob('/tmp/kt$PC/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the lookups-only phase: take a window of kernels between the 3rd and 4th-from... print the first 45 kernels after the 12th cache_tick
ticks=[i for i,r in enumerate(rows) if 'cache_tick' in r['Kernel_Name']]
a=ticks[11]; b=ticks[12]
t0=int(rows[a]['Start_Timestamp']); prev=t0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:8.1f} gap {(s-prev)/1e3:7.1f} dur {(e-s)/1e3:7.1f} q={r.get('Queue_Id','?')} {r['Kernel_Name'][:70]}")
    prev=max(prev,e)
P
done
