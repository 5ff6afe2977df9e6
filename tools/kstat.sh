#!/bin/bash
# usage: tools/kstat.sh <tag> [ENV=VAL ...]   -- kernel durations of tools/one_scan.py under rocprofv3 (tuning aid)
tag=$1; shift
out=gpurun_out/${KSTAT_ROUND:-r03}/kstat/$tag; mkdir -p $out
export TMPDIR=/tmp SW_REPS=${SW_REPS:-3}
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 ${KSTAT_TOOL:-tools/one_scan.py} > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
find $out -type f ! -name "*kernel_stats.csv" ! -name run.log -delete
python3 - $out "$tag" <<'PY'
import csv,sys,json,glob,os
out,tag=sys.argv[1],sys.argv[2]
f=glob.glob(out+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
keys=os.environ.get('KSTAT_KEYS','bucket_tag,bucket_index,part_scatter,sub_scatter,sub_probe,sub_index,verify_kernel,overflow_probe,hit_hist,group_scan,hit_partition_kernel<true>,hit_partition_kernel<false>,group_place,row_geo').split(',')
d={}
for r in rows:
    for k in keys:
        if k in r['Name']: d[k]=round(float(r['AverageNs'])/1e6,3)
js=[json.loads(l) for l in open(out+'/run.log') if l.startswith('{')]
print(tag, d, 'scan', [round(j['ms_scan'],2) for j in js][1:], 'chunks', js[0].get('part_chunks'))
PY
