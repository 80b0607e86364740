#!/bin/bash
# dev helper: A/B environment settings on the same box, alternating.  usage: ab.sh [repeats] "ENV_A=1" "ENV_B=1" ...
n=$1; shift
run() {
  out=$(env $1 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs 2>/dev/null | tail -1)
  echo "[$1] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['sweep_fwd_ms'],2))")"
}
for i in $(seq $n); do for v in "$@"; do run "$v"; done; done
