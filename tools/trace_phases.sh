#!/bin/bash
# dev helper: per-workgroup phase cycles (SMG_TRACE_KIND / SMG_TRACE_SKIP) for a few gemm_kernel launches
t() { echo -n "kind=$1 skip=$2: "; SMG_TRACE_KIND=$1 SMG_TRACE_SKIP=$2 timeout 120 python bench.py --steps 1 --warmup 0 --cpu-samples 0 --batched-scenes 0 --no-roofline 2>&1 | grep "smg trace" | head -1; }
for s in 0 5 17 18 41; do t 1 $s; done      # conv1x1 fwd: block1 layer 0 / 5, block2 layer 11, block3 layer 0 / 23
for s in 3 20; do t 7 $s; done
for s in 0 16; do t 8 $s; done              # conv1x1 wgrad
