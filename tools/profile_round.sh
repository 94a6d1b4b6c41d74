#!/bin/bash
# usage: tools/profile_round.sh <tag> [part] ; every measurement file of a round into gpurun_out/<tag>/ (copy what is to be
# judged into profiles/).  part 1: bench lines (headline, the driver's short form, the SURVEY 8d sub-runs: both antialias
# settings of the peripheral / flexible kinds, the packed ragged-raw form, gray screens) + rocprofv3 kernel-trace stats of the
# same commands; part 2: PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated) and the SQ / TCC counter
# summaries.  Default: both.  Progress goes to the log files as it runs (a silent command is killed after 7 minutes).
set -u
TAG=${1:-r04}
PART=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -x tools/membench ] || hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
python3 -c "import sys; sys.path.insert(0, 'active-gym_amd'); from active_gym import _native as n; print(n.build_info())" > $OUT/build.txt
SUB="peripheral:--kind@peripheral@--antialias@1 peripheral_aa0:--kind@peripheral@--antialias@0 flexible:--kind@flexible@--antialias@1 flexible_aa0:--kind@flexible@--antialias@0 flexible_packed:--kind@flexible@--out@packed flexible_packed_aa0:--kind@flexible@--out@packed@--antialias@0"
if [ "$PART" = all ] || [ "$PART" = 1 ]; then
  echo "== bench lines"
  python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $OUT/bench_driver_form.json 2>> $OUT/bench.err
  for s in $SUB; do
    name=${s%%:*}; args=$(echo ${s#*:} | tr '@' ' ')
    python3 bench.py $args --no-cpu-baseline --no-e2e > $OUT/bench_$name.json 2>> $OUT/bench.err
  done
  python3 bench.py --frame-format gray --no-cpu-baseline --no-e2e > $OUT/bench_gray_frames.json 2>> $OUT/bench.err
  echo "== kernel trace"
  for s in fixed: $SUB; do
    name=${s%%:*}; args=$(echo ${s#*:} | tr '@' ' ')
    # --obs-pool 0: the trace's average of the fovea kernel is then the product form alone (the pool pass has its own events)
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$name -- python3 bench.py $args --obs-pool 0 --no-cpu-baseline --no-e2e > $OUT/bench_under_rocprof_$name.json 2> $OUT/kt_$name.err || echo "kernel trace $name failed"
    f=$(find $OUT/kt_$name -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && { head -1 $f; grep 'agx::' $f; } > $OUT/kernel_stats_$name.csv
  done
fi
if [ "$PART" = all ] || [ "$PART" = 2 ]; then
  echo "== traffic"
  for k in fixed peripheral flexible; do
    AGX_TRAFFIC_BENCH_ARGS="--kind $k" bash tools/traffic.sh $OUT/traffic_$k > $OUT/traffic_$k.log 2>&1
    cp $OUT/traffic_$k/traffic.json $OUT/traffic_$k.json 2>/dev/null
  done
  # the same step on compact input screens (k_ingest_full12_compact): what K1 fetches when its rows are a gap-free stream
  AGX_TRAFFIC_BENCH_ARGS="--kind fixed --compact" bash tools/traffic.sh $OUT/traffic_fixed_compact > $OUT/traffic_fixed_compact.log 2>&1
  cp $OUT/traffic_fixed_compact/traffic.json $OUT/traffic_fixed_compact.json 2>/dev/null
  echo "== pmc"
  for k in fixed peripheral flexible; do
    AGX_PMC_BENCH_ARGS="--kind $k" bash tools/pmc.sh $OUT/pmc_$k > $OUT/pmc_$k.log 2>&1
    cp $OUT/pmc_$k/summary.txt $OUT/pmc_summary_$k.txt 2>/dev/null
  done
fi
ls $OUT | head -80
