#!/bin/bash
# dev helper: sweep the 3x3 weight-gradient workgroup targets, print per-stage ms
run() {
  out=$(env "$@" timeout 200 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --batched-scenes 0 2>/dev/null | tail -1)
  echo "$* $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['roofline']['per_stage']['conv3x3_wgrad'])")"
}
for w in 256 340 512 768 1024 1700 3400; do run SMG_W3_WGS16=$w; done
for w in 128 256 384 768 1700; do run SMG_W3_WGS8=$w; done
