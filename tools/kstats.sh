#!/bin/bash
# usage: tools/kstats.sh <outdir> [bench args...] ; rocprofv3 kernel-trace of a short bench run, prints agx kernel durations
OUT=${1:-gpurun_out/kstats}; shift
mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-events "$@" > $OUT/run.log 2>&1 || { echo "rocprof run failed"; tail -5 $OUT/run.log; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
d=collections.defaultdict(list); meta={}
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'agx::' not in k: continue
        d[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp'])); meta[k]=(r['VGPR_Count'],r['SGPR_Count'],r['LDS_Block_Size'],r['Grid_Size_X'],r['Grid_Size_Y'])
for k,v in d.items():
    v=sorted(v[10:]) if len(v)>20 else sorted(v)
    print('%-70s n=%d avg=%.2fus med=%.2f min=%.2f  vgpr,sgpr,lds,grid=%s'%(k[:70],len(v),sum(v)/len(v)/1e3,v[len(v)//2]/1e3,v[0]/1e3,meta[k]))
PY
