#!/bin/bash
# Evidence for one BASELINE.json side leg (config3 / config5) in one GPU call: serialised rocprofv3 kernel stats of the leg's training
# steps + HBM traffic (two separate --pmc passes).  usage (repo root, GPU box): tools/profile_leg.sh <leg> <commit> <round tag>
LEG=$1; COMMIT=${2:-unknown}; TAG=${3:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_$LEG; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --leg $LEG --train-only --steps 4 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- $CMD --serialize > $OUT/serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
cp "$(find $OUT/serial -name '*kernel_stats.csv' | head -1)" $OUT/rocprof_${TAG}_${LEG}_serialized.csv
echo "{\"commit\": \"$COMMIT\", \"command\": \"$CMD --serialize\", \"train_steps\": 5}" > $OUT/rocprof_${TAG}_${LEG}_serialized.meta.json
python3 tools/pmc_hbm_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_${TAG}_${LEG}_hbm_traffic "$COMMIT" "$CMD" > /dev/null
tail -2 $OUT/serial.log | cut -c1-400
