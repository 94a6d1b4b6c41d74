#!/bin/bash
# usage: tools/envab.sh "VAR=1" ["VAR2=3" ...] ; bench.py (300 steps, kernel events) with each environment setting, default first and last
set -u
B="python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-e2e"
sum() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
k=d['kernels']
print(sys.argv[2].ljust(24), '%.2fM %.2fus'%(d['value']/1e6, d['ms_per_step']*1e3), {a:round(b['avg_us'],2) for a,b in k.items()})
" $1 "$2"; }
mkdir -p gpurun_out/envab
$B ${BENCH_ARGS:-} > gpurun_out/envab/default_a.json 2>/dev/null; sum gpurun_out/envab/default_a.json default
i=0
for v in "$@"; do
  i=$((i+1))
  env $v $B ${BENCH_ARGS:-} > gpurun_out/envab/v$i.json 2>/dev/null; sum gpurun_out/envab/v$i.json "$v"
done
$B ${BENCH_ARGS:-} > gpurun_out/envab/default_b.json 2>/dev/null; sum gpurun_out/envab/default_b.json default
