#!/bin/bash
# dev helper: SQ counters per kernel (one rocprofv3 --pmc pass, kernel-trace only).  usage: tools/pmc_sq.sh [bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --train-only --steps 1 --warmup 1 "$@" > gpurun_out/pmc_sq.log 2>&1
python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq gpurun_out/pmc_sq_summary
