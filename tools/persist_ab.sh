#!/bin/bash
# usage: tools/persist_ab.sh ; same-box A/B of K1's resident-workgroup forms against the default launch (AGX_INGEST_STREAM=G)
set -u
mkdir -p gpurun_out/persist
B="python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-e2e"
sum() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
k=d['kernels']
print(sys.argv[1].split('/')[-1], '%.2fM %.2fus'%(d['value']/1e6, d['ms_per_step']*1e3), {a:round(b['avg_us'],2) for a,b in k.items()})
" $1; }
echo "== parity"
AGX_INGEST_STREAM=1792 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "ingest" > gpurun_out/persist/t_1792.log 2>&1; tail -2 gpurun_out/persist/t_1792.log
AGX_INGEST_STREAM=3 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "ingest" > gpurun_out/persist/t_3.log 2>&1; tail -2 gpurun_out/persist/t_3.log
AGX_LIB=$PWD/active-gym_amd/lib/libagx_s64.so AGX_INGEST_STREAM=5 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "ingest" > gpurun_out/persist/t_5_s64.log 2>&1; tail -2 gpurun_out/persist/t_5_s64.log
echo "== bench"
$B > gpurun_out/persist/b_default.json 2>/dev/null; sum gpurun_out/persist/b_default.json
for g in 1792 2048 1024 1536 3584; do
  AGX_INGEST_STREAM=$g $B > gpurun_out/persist/b_s66_$g.json 2>/dev/null; sum gpurun_out/persist/b_s66_$g.json
done
for g in 1792 2048 1024 3584; do
  AGX_LIB=$PWD/active-gym_amd/lib/libagx_s64.so AGX_INGEST_STREAM=$g $B > gpurun_out/persist/b_s64_$g.json 2>/dev/null; sum gpurun_out/persist/b_s64_$g.json
done
$B > gpurun_out/persist/b_default2.json 2>/dev/null; sum gpurun_out/persist/b_default2.json
