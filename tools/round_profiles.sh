#!/bin/bash
# Round artifacts on one MI355X: bench lines, rocprofv3 kernel statistics, steady-state per-step table, PMC HBM-traffic passes.
# Run from the repo root through gpurun; everything lands in gpurun_out/$TAG/ (copy what should be judged into profiles/).
#   bash tools/round_profiles.sh r03 [part]        part: all (default) | bench | prof | pmc | configs
TAG=${1:-r03}
PART=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
EAGER="--no-graph --no-overlap --no-cpu-baseline --no-roofline --no-drift --no-dropin"
set -x
if [ $PART = all ] || [ $PART = bench ]; then
  # 1. bench lines (HIP-graph replay, the driver's command), fp32 mode, the drop-in loop as its own run
  python3 bench.py --steps 20 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err || exit 1
  python3 bench.py --steps 20 --dtype f32 --no-cpu-baseline --no-dropin > $OUT/bench_f32.json 2> $OUT/bench_f32.err || exit 1
  python3 bench.py --steps 20 --mode dropin --no-cpu-baseline --no-drift > $OUT/bench_dropin.json 2> $OUT/bench_dropin.err || exit 1
fi
if [ $PART = all ] || [ $PART = configs ]; then
  python3 bench.py --steps 10 --designs 1 --nodes 300000 --levels 120 --tile 512 --batch-paths 4096 --no-cpu-baseline --no-drift --no-dropin \
    > $OUT/bench_configC.json 2> $OUT/bench_configC.err || exit 1
  python3 bench.py --steps 10 --designs 1 --nodes 1048576 --levels 128 --tile 512 --batch-paths 4096 --fanin irregular --no-cpu-baseline --no-drift --no-dropin \
    > $OUT/bench_configE.json 2> $OUT/bench_configE.err || exit 1
fi
if [ $PART = all ] || [ $PART = prof ]; then
  # 2. kernel statistics: the bench command itself (graph replay), the eager single-stream form at two step counts (their
  #    difference = the steady-state step), fp32 mode, the drop-in loop
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_graph -- python3 bench.py --steps 20 --no-cpu-baseline --no-drift --no-dropin > $OUT/prof_graph.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eager_bf16 -- python3 bench.py $EAGER --steps 10 > $OUT/prof_eager_bf16.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eager_bf16_30 -- python3 bench.py $EAGER --steps 30 > $OUT/prof_eager_bf16_30.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eager_f32 -- python3 bench.py $EAGER --steps 10 --dtype f32 > $OUT/prof_eager_f32.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dropin -- python3 bench.py $EAGER --steps 10 --mode dropin > $OUT/prof_dropin.log 2>&1 || exit 1
  for d in prof_graph prof_eager_bf16 prof_eager_bf16_30 prof_eager_f32 prof_dropin; do
    f=$(ls $OUT/$d/*/*_kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp "$f" $OUT/${d}_kernel_stats.csv
    python3 tools/trace_overlap.py $(ls $OUT/$d/*/*_kernel_trace.csv | head -1) > $OUT/${d}_overlap.txt 2>&1
    rm -rf $OUT/$d
  done
  python3 tools/steady_state_diff.py $OUT/prof_eager_bf16_kernel_stats.csv 10 $OUT/prof_eager_bf16_30_kernel_stats.csv 30 \
    $OUT/steady_state_per_step.csv > $OUT/steady_state_per_step.txt 2>&1
fi
if [ $PART = all ] || [ $PART = pmc ]; then
  # 3. HBM traffic: two separate counter passes (no tracing flags beside --pmc)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $EAGER --steps 4 > $OUT/pmc_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $EAGER --steps 4 > $OUT/pmc_write.log 2>&1 || exit 1
  python3 tools/pmc_to_json.py $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) $OUT/pmc_hbm_traffic.json 7
  rm -rf $OUT/pmc_fetch $OUT/pmc_write
fi
set +x
ls -la $OUT
