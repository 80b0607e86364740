#!/bin/bash
# dev helper: A/B environment settings on one bench leg, alternating.  usage: ab_leg.sh leg repeats "ENV_A=1" "ENV_B=1" ...
leg=$1; n=$2; shift 2
run() {
  out=$(env $1 timeout 300 python bench.py --leg $leg --steps 8 --warmup 2 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs 2>/dev/null | tail -1)
  echo "[$leg $1] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")"
}
for i in $(seq $n); do for v in "$@"; do run "$v"; done; done
