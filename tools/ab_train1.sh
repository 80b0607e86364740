#!/bin/bash
# single-sample step A/B: train_step_ms for each lib, alternating
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do for L in - smg-multimodal-grasping_amd/libsmg_base.so; do
  if [ "$L" != "-" ]; then export SMG_HIP_LIB=$GRAFT_REPO_ROOT/$L; else unset SMG_HIP_LIB; fi
  python bench.py --steps 10 --warmup 3 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],2), round(d['train_step_ms'],3), round(d['train_step_kernel_ms'],3))"
done; done
