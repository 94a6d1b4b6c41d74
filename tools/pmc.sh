#!/bin/bash
# usage: tools_pmc.sh <outdir> ; runs separate rocprofv3 --pmc passes over a short bench (counters only, no trace domains)
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-events ${AGX_PMC_BENCH_ARGS:-}"
i=0
for grp in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
  "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'agx::' not in k: continue
        k=k.split('(')[0].replace('void ','')[:110]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for k,d in agg.items():
        fo.write(k+'\n')
        for c,v in sorted(d.items()):
            fo.write('  %-32s n=%d mean=%.4g\n'%(c,len(v),sum(v)/len(v)))
print(open(out+'/summary.txt').read())
PY
