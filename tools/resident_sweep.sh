#!/bin/bash
run() {
  out=$(env "$@" timeout 200 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --batched-scenes 0 2>/dev/null | tail -1)
  echo "$* $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(round(d['ms_per_step'],2), round(d['sweep_fwd_ms'],2), {k: round(v['ms_per_step'],2) for k,v in r.items() if v['ms_per_step']>0.3})")"
}
run SMG_RESIDENT=0
run SMG_RESIDENT=1
run SMG_RESIDENT=2
run SMG_RESIDENT=3
run SMG_RESIDENT=4
run SMG_RESIDENT=1000
