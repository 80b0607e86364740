#!/bin/bash
# dev helper (review item 7, round 6): HBM traffic per kernel class with every launch SERIALISED in issue order (bench.py --serialize) -
# a dense layer's 1x1 weight gradient then runs directly in front of its 1x1 data gradient(s), both reading the layer's D2 units and
# the block's x columns: whatever the 256 MB MALL keeps between them shows as fewer fetched bytes than in the concurrent run.
# usage (repo root, GPU box): tools/pmc_serialized.sh <commit> <tag>
COMMIT=${1:-unknown}; TAG=${2:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_ser_$TAG; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --train-only --steps 4 --warmup 1 --serialize"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
python3 tools/pmc_hbm_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_${TAG}_serialized_hbm_traffic "$COMMIT" "$CMD"
cat $OUT/pmc_${TAG}_serialized_hbm_traffic.md
