#!/bin/bash
# single-sample step latency A/B: lib variants alternating
n=$1; shift
cd "$GRAFT_REPO_ROOT"
for i in $(seq $n); do for v in "$@"; do
  if [ "$v" != "-" ]; then export SMG_HIP_LIB=$GRAFT_REPO_ROOT/$v; else unset SMG_HIP_LIB; fi
  out=$(timeout 300 python bench.py --steps 20 --warmup 3 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs 2>/dev/null | tail -1)
  echo "[$v] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'step1', round(d['train_step_ms'],3), 'enq', round(d['train_step_host_enqueue_ms'],3), 'graph', round(d['train_step_graph_ms'],3), 'kern', round(d['train_step_kernel_ms'],3), d['train_step_launches'])")"
done; done
