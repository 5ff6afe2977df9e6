#!/bin/bash
# usage: tools/timeline.sh <tag> <python tool> [ENV=VAL ...] -- kernel timeline of the tool's LAST scan (tuning aid)
tag=$1; tool=$2; shift 2
out=gpurun_out/${KSTAT_ROUND:-r03}/timeline/$tag; mkdir -p $out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $tool > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 - $out <<'PY'
import csv,sys,glob
out=sys.argv[1]
f=glob.glob(out+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last scan = from the last build_blocks kernel on
idx=[i for i,r in enumerate(rows) if 'build_blocks' in r['Kernel_Name']]
rows=rows[idx[-1]:]
t0=int(rows[0]['Start_Timestamp'])
with open(out+'/timeline.txt','w') as fo:
    for r in rows:
        n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('kg::','')[:40]
        line="%8.1f %8.1f  q%-3s %s" % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Queue_Id','?'), n)
        fo.write(line+'\n')
print(open(out+'/timeline.txt').read())
PY
find $out -type f ! -name timeline.txt ! -name run.log -delete
