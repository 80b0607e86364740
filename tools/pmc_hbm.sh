#!/bin/bash
# HBM-side traffic per kernel class: two separate rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE; kernel-trace
# only), as MI355X_MICROARCH.md section HBM prescribes.  Writes gpurun_out/pmc_hbm_traffic.json / .md
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
CMD="python3 bench.py --steps 2 --warmup 1 --cpu-samples 0 --no-roofline --no-configs --batched-scenes 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- $CMD > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- $CMD > gpurun_out/pmc_write.log 2>&1
python3 tools/pmc_hbm_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_hbm_traffic
