#!/bin/bash
# dev helper: per-workgroup phase cycles (SMG_TRACE_KIND / SMG_TRACE_SKIP) for a few gemm_kernel launches.
# Kinds: 1 conv1x1 fwd, 7 conv1x1 dgrad, 8 conv1x1 wgrad (engine.hip K_*).  Launches are counted from process start.
# usage: trace_phases.sh kind "skip skip ..." [ENV=.. ...]
kind=$1; skips=$2; shift 2
t() { echo -n "kind=$kind skip=$1: "; env "${@:2}" SMG_TRACE_KIND=$kind SMG_TRACE_SKIP=$1 timeout 120 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs 2>&1 | grep "smg trace" | head -2; }
for s in $skips; do t $s "$@"; done
