#!/bin/bash
# usage: tools/sweep.sh [outdir] ; bench.py --envs N for N = 256 ... 8192 on one box (same call), one summary line per N
OUT=${1:-gpurun_out/sweep}; mkdir -p $OUT
for n in 256 512 1024 2048 4096 8192; do
  python3 bench.py --envs $n --steps 200 --warmup 20 --no-cpu-baseline --no-e2e > $OUT/sweep_$n.json 2>> $OUT/sweep.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); k=d['kernels']
print('N=%5d  %6.2f M env steps/s  %7.2f us/step  '%(int(sys.argv[2]), d['value']/1e6, d['ms_per_step']*1e3) + '  '.join('%s %.2f us (%.3f)'%(a, b['avg_us'], b['frac']) for a,b in k.items()))
" $OUT/sweep_$n.json $n
done
