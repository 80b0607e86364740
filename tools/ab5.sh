#!/bin/bash
# usage: ab5.sh [repeats] "ENV=.." ...
n=$1; shift
for i in $(seq $n); do for v in "$@"; do echo "[$v] $(env $v timeout 300 python tools/ab5.py 2>/dev/null | tail -1)"; done; done
