#!/bin/bash
# dev helper: A/B library builds on the same box, alternating (headline step + forward sweep).  usage: ab_lib.sh repeats lib1.so lib2.so ...  ("-" = the default build)
n=$1; shift
cd "$GRAFT_REPO_ROOT"
run() {
  if [ "$1" != "-" ]; then export SMG_HIP_LIB=$GRAFT_REPO_ROOT/$1; else unset SMG_HIP_LIB; fi
  out=$(timeout 300 python bench.py --steps 10 --warmup 3 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs 2>/dev/null | tail -1)
  echo "[$1] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d.get('sweep_fwd_ms',0),2), d.get('train_step_ms'))")"
}
for i in $(seq $n); do for v in "$@"; do run "$v"; done; done
