#!/bin/bash
# dev helper: per-workgroup phase cycles of one launch on a bench leg.  usage: trace_leg.sh leg kind "skip skip ..." [ENV=.. ...]
leg=$1; kind=$2; skips=$3; shift 3
for s in $skips; do echo -n "leg=$leg kind=$kind skip=$s: "; env "$@" SMG_TRACE_KIND=$kind SMG_TRACE_SKIP=$s timeout 300 python bench.py --leg $leg --steps 2 --warmup 1 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs 2>&1 | grep "smg trace" | head -1; done
