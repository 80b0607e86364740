#!/bin/bash
# Round evidence in one GPU call (run from the repo root on the GPU box: tools/profile_round.sh <commit> <round tag>):
#   1. rocprofv3 --kernel-trace --stats of the training step as it runs (two concurrent chains)
#   2. the same with every launch serialised on one stream (bench.py --serialize): per-kernel durations a reader can
#      recompute bench.py's roofline fractions from
#   3. HBM traffic: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only
# Everything is a `bench.py --train-only` run: every kernel in a trace belongs to a training step.
COMMIT=${1:-unknown}; TAG=${2:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --train-only --steps 4 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/conc -- $CMD > $OUT/conc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- $CMD --serialize > $OUT/serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
cp "$(find $OUT/conc -name '*kernel_stats.csv' | head -1)" $OUT/rocprof_${TAG}_kernel_stats.csv
cp "$(find $OUT/serial -name '*kernel_stats.csv' | head -1)" $OUT/rocprof_${TAG}_kernel_stats_serialized.csv
echo "{\"commit\": \"$COMMIT\", \"command\": \"$CMD --serialize\", \"train_steps\": 5}" > $OUT/rocprof_${TAG}_kernel_stats_serialized.meta.json
python3 tools/pmc_hbm_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_${TAG}_hbm_traffic "$COMMIT" "$CMD"
python3 tools/timeline.py "$(find $OUT/conc -name '*kernel_trace.csv' | head -1)" > $OUT/timeline_${TAG}.txt 2>&1
tail -3 $OUT/conc.log $OUT/serial.log
head -30 $OUT/rocprof_${TAG}_kernel_stats_serialized.csv | cut -c1-200
