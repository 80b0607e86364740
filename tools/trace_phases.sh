#!/bin/bash
# dev helper: per-workgroup phase cycles (SMG_TRACE_KIND / SMG_TRACE_SKIP) for a few gemm_kernel launches.
# Kinds: 1 conv1x1 fwd, 7 conv1x1 dgrad, 8 conv1x1 wgrad (engine.hip K_*).  Launches are counted from process start; the
# forward issues 58 conv1x1 launches per chain-less pass, so skip >= 116 lands in the third (warm) forward.
t() { echo -n "kind=$1 skip=$2: "; SMG_FWD_ONE_CHAIN=1 SMG_TRACE_KIND=$1 SMG_TRACE_SKIP=$2 timeout 120 python bench.py --steps 2 --warmup 1 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs 2>&1 | grep "smg trace" | head -1; }
for s in 116 121 133 134 145 157 173; do t 1 $s; done
