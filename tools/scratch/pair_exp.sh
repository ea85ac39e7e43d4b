#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  tag=$1; shift
  OUT=gpurun_out/pairexp_$tag; mkdir -p $OUT
  env "$@" python3 tools/scratch/pair_exp_run.py $OUT > $OUT/run.log 2>&1
  echo "== $tag: $(tail -1 $OUT/run.log)"
}
run base_p16 MMFT_PAIR_PART=16
run base_p32 MMFT_PAIR_PART=32
run s48_p48 MMFT_PAIR_SINKS=48 MMFT_PAIR_PART=48
run nofence MMFT_PAIR_DBG=1
