#!/bin/bash
# usage: tools/ab2.sh name... ; bench.py (300 steps, kernel events) with active-gym_amd/lib/libagx_<name>.so, default library first and last
set -u
B="python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-e2e ${BENCH_ARGS:-}"
sum() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
k=d['kernels']
print(sys.argv[1].split('/')[-1], '%.2fM %.2fus'%(d['value']/1e6, d['ms_per_step']*1e3), {a:round(b['avg_us'],2) for a,b in k.items()})
" $1; }
mkdir -p gpurun_out/ab2
$B > gpurun_out/ab2/default_a.json 2>/dev/null; sum gpurun_out/ab2/default_a.json
for v in "$@"; do
  AGX_LIB=$PWD/active-gym_amd/lib/libagx_$v.so $B > gpurun_out/ab2/$v.json 2>/dev/null; sum gpurun_out/ab2/$v.json
done
$B > gpurun_out/ab2/default_b.json 2>/dev/null; sum gpurun_out/ab2/default_b.json
