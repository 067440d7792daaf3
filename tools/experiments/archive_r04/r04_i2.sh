#!/bin/bash
# kernel-trace timeline of the training step: per-queue busy time, gaps, overlap
cd "$(dirname "$0")/../.." || exit 1
root=$(pwd); O=gpurun_out/r04i2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/$O/prof -- python3 $root/bench.py --workload train --no-cpu-baseline --no-kernel-profile --steps 6 --warmup 3 > /dev/null 2>&1
cd $root
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/timeline.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the last 3 steps: split by sgd kernel occurrences
idx=[i for i,r in enumerate(rows) if 'sgd_nesterov' in r['Kernel_Name']]
# steps end at every 3rd sgd call
ends=idx[2::3]
a,b=ends[-3],ends[-2]   # one full step between two optimizer ends
step=rows[a+1:b+1]
t0=int(step[0]['Start_Timestamp']); t1=int(step[-1]['End_Timestamp'])
print("step wall us", (t1-t0)/1e3, "kernels", len(step))
byq=collections.defaultdict(list)
for r in step: byq[r['Queue_Id']].append(r)
for q,rs in byq.items():
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rs)/1e3
    first=(int(rs[0]['Start_Timestamp'])-t0)/1e3; last=(int(rs[-1]['End_Timestamp'])-t0)/1e3
    gaps=sum(max(0,int(rs[i+1]['Start_Timestamp'])-int(rs[i]['End_Timestamp'])) for i in range(len(rs)-1))/1e3
    print(f"queue {q}: {len(rs)} kernels, busy {busy:.0f} us, span {first:.0f}..{last:.0f} us, gaps inside {gaps:.0f} us")
# union busy time of all queues and per-ms profile of which queue is active
ev=[]
for r in step: ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort(); cur=0; last=t0; idle=0; one=0; two=0
for t,d in ev:
    dt=t-last
    if cur==0: idle+=dt
    elif cur==1: one+=dt
    else: two+=dt
    cur+=d; last=t
print(f"GPU idle {idle/1e3:.0f} us, exactly one kernel {one/1e3:.0f} us, two or more {two/1e3:.0f} us")
# main queue = the one with most kernels; list its 25 largest gaps with neighbours
mq=max(byq,key=lambda q:len(byq[q])); rs=byq[mq]
g=[(int(rs[i+1]['Start_Timestamp'])-int(rs[i]['End_Timestamp']),rs[i]['Kernel_Name'][:50],rs[i+1]['Kernel_Name'][:50],(int(rs[i]['End_Timestamp'])-t0)/1e3) for i in range(len(rs)-1)]
g.sort(reverse=True)
for x in g[:15]: print(f"gap {x[0]/1e3:7.1f} us at {x[3]:8.0f}: {x[1]} -> {x[2]}")
# the step boundary in detail: every dispatch (both queues) in the last 700 us of the step and the first 700 us of the next one
nxt=rows[b+1:b+40]
print("--- boundary (times relative to the step start of the FOLLOWING step) ---")
tb=int(nxt[0]['Start_Timestamp']) if nxt else t1
for r in step[-14:]+nxt[:16]:
    print(f"q{r['Queue_Id']} {(int(r['Start_Timestamp'])-tb)/1e3:9.1f} .. {(int(r['End_Timestamp'])-tb)/1e3:9.1f}  {r['Kernel_Name'][:70]}")
PY
rm -rf $O/prof
