R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c2f16_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c2f16_prof -- python3 $R/tools/experiments/r06_c2f16_time.py > /dev/null 2>&1
cd $R
f=$(find gpurun_out/c2f16_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'c2f16' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3)
PY
f2=$(find gpurun_out/c2f16_prof -name "*kernel_trace.csv" | head -1)
python3 - "$f2" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'c2f16_stream' in r['Kernel_Name']]
for r in rows: r['Grid_Size']=r['Kernel_Name'][:22]+' '+r.get('Grid_Size','')
# group consecutive runs of 23 launches (one per rows setting) by grid size
g=collections.OrderedDict()
for r in rows:
    k=(r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'), )
    g.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in g.items(): print('grid',k,'n',len(v),'median us',sorted(v)[len(v)//2])
PY
rm -rf gpurun_out/c2f16_prof
