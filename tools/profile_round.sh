#!/bin/bash
# usage: tools/profile_round.sh <tag> ; every measurement file of a round into gpurun_out/<tag>/ (copy what is to be
# judged into profiles/): bench lines (fixed / peripheral / flexible / gray frames), rocprofv3 kernel-trace stats of the
# same commands, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated) and the SQ / TCC counter summaries.
# Progress goes to the log files as it runs (a silent command is killed after 7 minutes).
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -x tools/membench ] || hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
python3 -c "import sys; sys.path.insert(0, 'active-gym_amd'); from active_gym import _native as n; print(n.build_info())" > $OUT/build.txt
echo "== bench lines"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
for k in peripheral flexible; do python3 bench.py --kind $k --no-cpu-baseline --no-e2e > $OUT/bench_$k.json 2>> $OUT/bench.err; done
python3 bench.py --frame-format gray --no-cpu-baseline --no-e2e > $OUT/bench_gray_frames.json 2>> $OUT/bench.err
echo "== kernel trace"
for k in fixed peripheral flexible; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$k -- python3 bench.py --kind $k --no-cpu-baseline --no-e2e > $OUT/bench_under_rocprof_$k.json 2> $OUT/kt_$k.err || echo "kernel trace $k failed"
  f=$(find $OUT/kt_$k -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { head -1 $f; grep 'agx::' $f; } > $OUT/kernel_stats_$k.csv
done
echo "== traffic"
for k in fixed peripheral flexible; do
  AGX_TRAFFIC_BENCH_ARGS="--kind $k" bash tools/traffic.sh $OUT/traffic_$k > $OUT/traffic_$k.log 2>&1
  cp $OUT/traffic_$k/traffic.json $OUT/traffic_$k.json 2>/dev/null
done
echo "== pmc"
for k in fixed peripheral flexible; do
  AGX_PMC_BENCH_ARGS="--kind $k" bash tools/pmc.sh $OUT/pmc_$k > $OUT/pmc_$k.log 2>&1
  cp $OUT/pmc_$k/summary.txt $OUT/pmc_summary_$k.txt 2>/dev/null
done
ls $OUT | head -60
