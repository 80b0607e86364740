#!/bin/bash
# dev helper: sweep the conv1x1 weight-gradient launch knobs (engine.hip SMG_W1_*), print per-stage ms
run() {
  out=$(env "$@" timeout 200 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --batched-scenes 0 2>/dev/null | tail -1)
  echo "$* $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']; print(round(d['ms_per_step'],2), d['roofline']['per_stage']['conv1x1_wgrad'], 'TW', round(r['transition_wgrad']['ms_per_step'],3), 'SW', round(r['stem_wgrad']['ms_per_step'],3))")"
}
for cfg in 0 1 2 3 4; do
  for wgs in 384 512 768; do
    run SMG_W1_CFG=$cfg SMG_W1_WGS=$wgs
  done
done
